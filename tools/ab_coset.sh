for i in 1 2; do
for v in 0 1; do
  echo "coset=$v"; ZKHIP_COSET_QUOTIENT=$v python bench.py --config rsa17 --no-other-configs --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readlines()[-1]); print(j['ms_per_step'], {k:v['ms_per_step'] for k,v in j['configs']['rsa17']['kernels_ms_per_step'].items()})"
done; done
ZKHIP_HOST_TIMING=1 ZKHIP_COSET_QUOTIENT=1 python bench.py --config rsa17 --no-other-configs --no-cpu-baseline --steps 3 --warmup 2 2>&1 | grep -v "^{" | tail -40
