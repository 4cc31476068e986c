# usage: bash tools/ab_env.sh <config> <steps> VAR=a VAR=b ...   — one bench run per setting, same box
cfg=$1; steps=$2; shift 2
for setting in "$@"; do
  echo "== $setting"
  env $setting python bench.py --config $cfg --no-other-configs --no-cpu-baseline --steps $steps --warmup 3 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readlines()[-1]); c=j['configs']['$cfg']; print(j['ms_per_step'], {k:v['ms_per_step'] for k,v in c['kernels_ms_per_step'].items() if v['ms_per_step']>0.5})"
done
