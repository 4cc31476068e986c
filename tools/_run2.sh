set -x
mkdir -p gpurun_out/r3b
python tools/setup_breakdown.py agg22 > gpurun_out/r3b/setup_agg22.txt 2>&1; cat gpurun_out/r3b/setup_agg22.txt
timeout 3000 python -m pytest tests/test_gpu_shplonk.py tests/test_gpu_msm.py tests/test_gpu_prover.py tests/test_gpu_bench_contract.py -x -q 2>&1 | tail -15
timeout 3000 python -m pytest tests/test_gpu_distributed.py -x -q -k "k22 or k19" 2>&1 | tail -15
