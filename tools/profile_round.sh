#!/bin/bash
# Collects the artefacts of one profiles/ generation on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r02_v1  -> gpurun_out/r02_v1/{bench.json, kernel_bench.txt, <cfg>_kernel_stats.csv, pmc_<cfg>.csv, valu_<cfg>.csv}
# rocprofv3 gets the program itself after `--`; counter passes are separate runs (no trace domains beside --pmc).
# Copy what is to be judged into profiles/ as <tag>_<file> (pmc files keep their `# build=<hash>` header: bench.py quotes the
# traffic figures only for the build they were measured on).
set -u
tag=${1:-prof}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
python3 tools/kernel_bench.py > "$out/kernel_bench.txt" 2> "$out/bench.err"
cd /tmp
for cfg in rsa17 sha19 agg22; do
    args="--config $cfg --no-other-configs --no-cpu-baseline --no-h2d --steps 3 --warmup 1"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_$cfg" -- python3 "$root/bench.py" $args > "$out/stats_$cfg.json" 2> "$out/stats_$cfg.err"
    for c in FETCH_SIZE WRITE_SIZE SQ_LDS_BANK_CONFLICT; do
        rocprofv3 --pmc $c --output-format csv -d "$out/pmc_${cfg}_$c" -- python3 "$root/bench.py" $args > "$out/pmc_${cfg}_$c.json" 2> "$out/pmc_${cfg}_$c.err"
    done
    i=0
    for set in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM"; do
        i=$((i+1))
        rocprofv3 --pmc $set --output-format csv -d "$out/valu_${cfg}_$i" -- python3 "$root/bench.py" $args > "$out/valu_${cfg}_$i.json" 2> "$out/valu_${cfg}_$i.err"
    done
done
cd "$root"
python3 tools/summarize_profiles.py "$out"
# the bench line last, with this generation's PMC passes in place: bench.py reads roofline.traffic from profiles/*_pmc_<cfg>.csv of the
# build it runs (matching `# build=` hash)
for cfg in rsa17 sha19 agg22; do cp "$out/pmc_$cfg.csv" "profiles/${tag}_pmc_$cfg.csv"; done
python3 bench.py --detail-out "$out/bench_detail.json" > "$out/bench.json" 2>> "$out/bench.err"
