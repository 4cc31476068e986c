#!/bin/bash
# round 6, second call: the pipelined host-slice MSM (parity + timing by chunk count), the bench line with ffi levels, the comm trace in a replay run
out=gpurun_out/r06b
mkdir -p $out
python -m pytest tests/test_gpu_msm.py -q -x -k "pipelined or ragged or trapdoor" > $out/msm_tests.txt 2>&1; tail -3 $out/msm_tests.txt
python tools/boundary_bench.py --k 20 22 --chunks 1 2 4 8 > $out/boundary_chunks.json 2> $out/boundary_chunks.err; cat $out/boundary_chunks.json
python tools/boundary_bench.py --k 17 22 > $out/boundary_after.json 2> $out/boundary_after.err
python bench.py --steps 5 --warmup 2 --detail-out $out/bench_detail.json > $out/bench.line 2> $out/bench.err; cat $out/bench.line; wc -c $out/bench.line
python bench.py --steps 4 --warmup 1 --no-other-configs --no-cpu-baseline --replay-rank 3 --of 8 --replay-latency-us 20 --replay-link-gbs 50 --detail-out $out/replay_detail.json > $out/replay.line 2> $out/replay.err; cat $out/replay.line
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06b/replay_detail.json"))
tr=d["replay"]["trace"]
print("trace passes", len(tr), "exchanges", len(tr[0]["done_us"]), "end_us", [p["end_us"] for p in tr])
for e,t in zip(tr[0]["exchanges"], tr[0]["done_us"]): print(e, t)
d=json.load(open("gpurun_out/r06b/bench_detail.json"))
print(json.dumps(d["ffi_levels"], indent=1))
print(json.dumps(d["configs"]["rsa17"].get("ffi_levels"), indent=1))
PY
