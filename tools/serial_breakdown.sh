# kernel-by-kernel cost of one proof with the overlap switched off (late_overlap = 2: everything on one stream), from a kernel trace
cfg=${1:-agg22}
export TMPDIR=/tmp
out=$(pwd)/gpurun_out/serial_$cfg
root=$(pwd)
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d "$out" -- python3 "$root/tools/trace_one.py" --config $cfg --steps 3 --set late_overlap=2 > "$out.log" 2>&1
cd "$root"
tail -1 "$out.log"
python3 tools/trace_gaps.py $(find "$out" -name "*kernel_trace.csv") | head -45
