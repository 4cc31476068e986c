#!/bin/bash
# Does any consumer on a proof's main stream lack its dependency on the side / third stream (or overwrite a buffer such a section still reads)?  ZKHIP_DEBUG_DELAY_US holds
# every such section back by that many microseconds (prover.hip k_debug_delay): the chain's five proof digests and the prover parity file must not notice.
#   gpurun -- bash tools/delay_check.sh
out=gpurun_out/r06_delay
mkdir -p $out
python bench.py --chain --steps 1 --warmup 1 --no-cpu-baseline --detail-out $out/chain_clean.json > /dev/null 2> $out/chain_clean.err
for us in 500 5000 30000; do
  ZKHIP_DEBUG_DELAY_US=$us python bench.py --chain --steps 1 --warmup 1 --no-cpu-baseline --detail-out $out/chain_$us.json > /dev/null 2> $out/chain_$us.err
  python - $us <<'PY'
import json, sys
us = sys.argv[1]
a = json.load(open("gpurun_out/r06_delay/chain_clean.json")); b = json.load(open(f"gpurun_out/r06_delay/chain_{us}.json"))
print(f"delay {us} us: chain digests equal: {a['proof_sha256'] == b['proof_sha256']}; step {a['ms_per_step']} -> {b['ms_per_step']} ms")
PY
done
for us in 5000 30000; do      # the reverse: the main stream held back behind every side section it has issued
  ZKHIP_DEBUG_DELAY_MAIN_US=$us python bench.py --chain --steps 1 --warmup 1 --no-cpu-baseline --detail-out $out/chain_main_$us.json > /dev/null 2> $out/chain_main_$us.err
  python - $us <<'PY'
import json, sys
us = sys.argv[1]
a = json.load(open("gpurun_out/r06_delay/chain_clean.json")); b = json.load(open(f"gpurun_out/r06_delay/chain_main_{us}.json"))
print(f"MAIN stream delayed {us} us: chain digests equal: {a['proof_sha256'] == b['proof_sha256']}; step {a['ms_per_step']} -> {b['ms_per_step']} ms")
PY
done
ZKHIP_DEBUG_DELAY_MAIN_US=3000 python -m pytest tests/test_gpu_prover.py -q -m gpu > $out/pytest_delay_main.txt 2>&1; tail -2 $out/pytest_delay_main.txt
ZKHIP_DEBUG_DELAY_US=3000 python -m pytest tests/test_gpu_prover.py tests/test_gpu_shplonk.py -q -m gpu > $out/pytest_delay.txt 2>&1; tail -3 $out/pytest_delay.txt
ZKHIP_DEBUG_DELAY_US=3000 python -m pytest tests/test_gpu_distributed.py -q -m gpu -k "sharded_proof_equals or k22_proof_over_two_ranks_by_point or sha_k19 or eight_and_five" > $out/pytest_delay_dist.txt 2>&1; tail -3 $out/pytest_delay_dist.txt
