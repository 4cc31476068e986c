#!/bin/bash
# The k = 22 aggregation-shaped proof over EIGHT ranks on one device (host-staged transport) against the single-GPU proof's bytes:
# BASELINE configs[3]'s rank count at full size, possible since round 4 (a rank of the sharded proof holds ~22 GiB).  Run through gpurun.
out=gpurun_out/${1:-n8}
mkdir -p $out
python bench.py --no-other-configs --no-cpu-baseline --no-h2d --steps 3 --warmup 1 > $out/n1.json 2> $out/n1.err
ZKHIP_BENCH_ONE_DEVICE=1 ZKHIP_BENCH_DIST_BACKEND=gloo timeout 1500 python bench.py --gpus 8 --steps 2 --warmup 1 --no-cpu-baseline --rung-budget 1200 --no-ladder > $out/n8.json 2> $out/n8.err
python - <<PY
import json
a = json.loads([l for l in open("$out/n1.json") if l.startswith("{")][-1])
print("N=1", a["value"], a["configs"]["agg22"]["proof_sha256"], round(a["resident_bytes"] / 2**30, 1), "GiB")
try:
    b = json.loads([l for l in open("$out/n8.json") if l.startswith("{")][-1])
    c = b["configs"]["agg22"]
    print("N=8 on one device", b["value"], c["proof_sha256"], round(c["resident_bytes"] / 2**30, 1), "GiB per rank", b["comm"], b["ladder"]["rung"])
    print("SAME BYTES" if c["proof_sha256"] == a["configs"]["agg22"]["proof_sha256"] else "DIFFERENT BYTES")
except Exception as e:
    print("N=8 failed", e)
PY
tail -5 $out/n8.err
