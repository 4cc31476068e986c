#!/bin/bash
# VALU-utilisation evidence for the integer-bound claim: SQ counter passes per bench configuration.
#   tools/profile_valu.sh r02_v1   -> gpurun_out/r02_v1/valu_<cfg>.csv (per-kernel totals of each counter)
# Counter passes are separate rocprofv3 runs with the program itself after `--` and no trace domains beside --pmc.
set -u
tag=${1:-valu}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > "$out/counters_list.txt" 2>&1 || true
run() {   # name, bench args...
    name=$1; shift
    for set in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM"; do
        d="$out/pmc_${name}_$(echo $set | tr ' ' '_' | cut -c1-40)"
        rocprofv3 --pmc $set --output-format csv -d "$d" -- python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs "$@" > "$d.json" 2> "$d.err"
    done
}
run k17
run sha19 --k 19 --shape sha256
run k22 --k 22
cd "$root"
python3 tools/summarize_valu.py "$out"
