timeout 1200 python -m pytest tests/test_gpu_msm.py -q -x 2>&1 | tail -3
for c in agg22 sha19 rsa17; do
python bench.py --config $c --steps 8 --warmup 2 --no-other-configs --no-cpu-baseline --no-h2d 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
c=d['configs'][d['config']['headline']]
print(d['config']['headline'],'ms_per_step',d['ms_per_step'],'digits',c['kernels_ms_per_step']['msm_digits'], 'accum', c['kernels_ms_per_step']['msm_accum_affine'],'tail',c['kernels_ms_per_step']['msm_tail'])
"
done
