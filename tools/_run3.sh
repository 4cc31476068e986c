mkdir -p gpurun_out/r3c
for l in 0 1 2 3; do
  echo "=== microbench opq$l"; ./tools/microbench_opq$l 2>&1 | grep "waves/CU=32" | grep -E "fe_mul|madd"
done 2>&1 | tee gpurun_out/r3c/microbench_opq.txt
for l in 0 1 2 3; do
  echo "=== bench opq$l"
  ZKHIP_LIB=$PWD/tools/ab/libzkhip_opq$l.so python bench.py --steps 8 --warmup 2 --no-other-configs --no-cpu-baseline --no-h2d 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
c=d['configs']['agg22']
print('ms_per_step',d['ms_per_step'],'accum launch ms',d['roofline']['avg_launch_ms'],'kernels',{k:v['ms_per_step'] for k,v in c['kernels_ms_per_step'].items() if v['ms_per_step']>1}, 'ntt iso', c['rooflines']['ntt']['achieved'])
"
  ZKHIP_LIB=$PWD/tools/ab/libzkhip_opq$l.so python bench.py --config rsa17 --steps 10 --warmup 2 --no-other-configs --no-cpu-baseline --no-h2d 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rsa17 ms_per_step',d['ms_per_step'])
"
done 2>&1 | tee gpurun_out/r3c/bench_opq.txt
