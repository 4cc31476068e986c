timeout 2400 python -m pytest tests/test_gpu_msm.py tests/test_gpu_polyops.py tests/test_gpu_prover.py tests/test_gpu_distributed.py -q -x 2>&1 | tail -5
for c in agg22 sha19 rsa17; do
python bench.py --config $c --steps 8 --warmup 2 --no-other-configs --no-cpu-baseline --no-h2d 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
c=d['configs'][d['config']['headline']]
print(d['config']['headline'],'ms_per_step',d['ms_per_step'],'accum launch ms',d['roofline']['avg_launch_ms'],'first_proof_s',d['first_proof_s'],'kernels',{k:v['ms_per_step'] for k,v in c['kernels_ms_per_step'].items() if v['ms_per_step']>0.3}, 'ntt iso', c['rooflines']['ntt']['achieved'])
"
done
