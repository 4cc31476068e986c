#!/bin/bash
# round 6, first measurement call: transfer probe, the boundary's host-pointer entry points as they are, contention determinism, the new RCCL self-test
out=gpurun_out/r06a
mkdir -p $out
./tools/h2d_probe 128 8 > $out/h2d_probe.txt 2>&1
python tools/boundary_bench.py --k 17 22 > $out/boundary_before.json 2> $out/boundary_before.err
python -m pytest tests/test_gpu_distributed.py -q -x -k "real_librccl or stuck_collective or single_rank" -s > $out/rccl_selftest.txt 2>&1
python tools/contend_stress.py --seconds 45 > $out/contend_base.json 2> $out/contend_base.err
python tools/contend_stress.py --seconds 45 --second-context > $out/contend_2ctx.json 2> $out/contend_2ctx.err
python tools/contend_stress.py --seconds 45 --env ZKHIP_LATE_OVERLAP=2 > $out/contend_serial.json 2> $out/contend_serial.err
for i in 1 2 3; do python -m pytest tests/test_gpu_bench_contract.py -q -x -k chain_six > $out/chain_six_$i.txt 2>&1; tail -1 $out/chain_six_$i.txt; done
tail -5 $out/h2d_probe.txt; cat $out/boundary_before.json | head -60; tail -5 $out/rccl_selftest.txt; cat $out/contend_*.json
