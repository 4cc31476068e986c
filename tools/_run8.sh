timeout 2400 python -m pytest tests/test_gpu_ntt.py tests/test_gpu_cosets.py tests/test_gpu_prover.py -q -x 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -8
python tools/ntt_plan_bench.py 2>&1 | tail -12
python bench.py --steps 8 --warmup 2 --no-other-configs --no-cpu-baseline --no-h2d 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
c=d['configs'][d['config']['headline']]
print(d['config']['headline'],'ms_per_step',d['ms_per_step'],'kernels',{k:v['ms_per_step'] for k,v in c['kernels_ms_per_step'].items() if v['ms_per_step']>0.3}, 'ntt iso', c['rooflines']['ntt']['achieved'])
"
