#!/bin/bash
# Collects the artefacts of one profiles/ generation on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r01_v8      -> gpurun_out/r01_v8/{bench*.json, kernel_stats.csv, pmc_fetch_write.csv, kernel_bench.txt}
# rocprofv3 gets the program itself after `--` and the counter passes are separate runs (no trace domains with --pmc).
set -u
tag=${1:-prof}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
python3 bench.py > "$out/bench.json" 2> "$out/bench.err"
python3 bench.py --k 22 --no-other-configs --steps 5 --no-cpu-baseline > "$out/bench_k22.json" 2>> "$out/bench.err"
python3 bench.py --k 19 --shape sha256 --no-other-configs --steps 10 --no-cpu-baseline > "$out/bench_sha256_k19.json" 2>> "$out/bench.err"
python3 tools/kernel_bench.py > "$out/kernel_bench.txt" 2>> "$out/bench.err"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 "$root/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-other-configs > "$out/stats_bench.json" 2> "$out/stats.err"
for c in FETCH_SIZE WRITE_SIZE SQ_LDS_BANK_CONFLICT; do
    rocprofv3 --pmc $c --output-format csv -d "$out/pmc_$c" -- python3 "$root/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > "$out/pmc_$c.json" 2> "$out/pmc_$c.err"
done
cd "$root"
python3 tools/summarize_profiles.py "$out"
