mkdir -p gpurun_out/r3d
timeout 1500 python -m pytest tests/test_gpu_msm.py tests/test_gpu_ntt.py tests/test_gpu_sweep.py -x -q 2>&1 | tail -4
for lib in tools/ab/libzkhip_opq0.so halo2-zkcert_amd/libzkhip.so; do
  echo "=== bench $lib"
  for c in agg22 sha19 rsa17; do
  ZKHIP_LIB=$PWD/$lib python bench.py --config $c --steps 8 --warmup 2 --no-other-configs --no-cpu-baseline --no-h2d 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
c=d['configs'][d['config']['headline']]
print(d['config']['headline'],'ms_per_step',d['ms_per_step'],'accum launch ms',d['roofline']['avg_launch_ms'],'kernels',{k:v['ms_per_step'] for k,v in c['kernels_ms_per_step'].items() if v['ms_per_step']>0.3}, 'ntt iso', c['rooflines']['ntt']['achieved'])
"
  done
done 2>&1 | tee gpurun_out/r3d/bench_ab.txt
