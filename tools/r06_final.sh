#!/bin/bash
# round 6, final verification on one box: smoke, the GPU suite with durations, the driver's bench command
out=gpurun_out/r06_final
mkdir -p $out
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; tail -1 $out/smoke.txt
( time python -m pytest tests -x -q -m gpu --durations=30 ) > $out/gpu_suite.txt 2>&1; tail -42 $out/gpu_suite.txt
python bench.py --gpus 1 --steps 20 --warmup 5 --detail-out $out/bench_detail.json > $out/bench.line 2> $out/bench.err; cat $out/bench.line; wc -c $out/bench.line
