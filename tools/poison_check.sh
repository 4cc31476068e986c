#!/bin/bash
# Does any kernel depend on a fresh allocation being zero?  ZKHIP_POISON=1 fills every fresh device allocation of the library (and every ffi.Context.empty tensor) with garbage:
# the single-GPU chain's five proof digests and the single-process parity files must not notice.
#   gpurun -- bash tools/poison_check.sh
out=gpurun_out/r06_poison
mkdir -p $out
python bench.py --chain --steps 1 --warmup 1 --no-cpu-baseline --detail-out $out/chain_clean.json > /dev/null 2> $out/chain_clean.err
ZKHIP_POISON=1 python bench.py --chain --steps 1 --warmup 1 --no-cpu-baseline --detail-out $out/chain_poison.json > /dev/null 2> $out/chain_poison.err
python - <<'PY'
import json
a=json.load(open("gpurun_out/r06_poison/chain_clean.json"))["proof_sha256"]
b=json.load(open("gpurun_out/r06_poison/chain_poison.json"))["proof_sha256"]
print("chain digests equal under poison:", a == b)
for x, y in zip(a, b): print("  ", x[:16], y[:16], x == y)
PY
ZKHIP_POISON=1 python -m pytest tests/test_gpu_prover.py tests/test_gpu_msm.py tests/test_gpu_ntt.py tests/test_gpu_cosets.py tests/test_gpu_polyops.py tests/test_gpu_shplonk.py tests/test_gpu_sweep.py -q -m gpu > $out/pytest_poison.txt 2>&1; tail -15 $out/pytest_poison.txt
ZKHIP_POISON=1 python -m pytest tests/test_gpu_distributed.py -q -m gpu -k "sharded_proof_equals or eight_and_five or k22_proof_over_two_ranks_by_point or sha_k19" > $out/pytest_poison_dist.txt 2>&1; tail -8 $out/pytest_poison_dist.txt
