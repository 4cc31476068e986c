set -x
mkdir -p gpurun_out/r3a
./tools/fp64_field > gpurun_out/r3a/fp64_field.txt 2>&1; tail -8 gpurun_out/r3a/fp64_field.txt
python tools/setup_breakdown.py agg22 > gpurun_out/r3a/setup_agg22.txt 2>&1; cat gpurun_out/r3a/setup_agg22.txt
timeout 1500 python -m pytest tests/test_gpu_bench_contract.py -x -q 2>&1 | tail -15
