#!/bin/bash
# One GPU-box pass of the round's checks (run through gpurun from the repo root): the -m gpu suite, then one profiles/ generation.
#   tools/run_round_checks.sh r03_v1
tag=${1:-prof}
timeout 3400 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -6
bash tools/profile_round.sh "$tag"
tail -c 400 "gpurun_out/$tag/bench.err"
