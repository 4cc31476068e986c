#!/usr/bin/env python3
"""profiles/r04_cpu_k22.json from the line of `python bench.py --cpu-baseline-k 22 --no-other-configs` (real CPU-oracle passes of the headline
shape at k = 18, 20 and 22 in ONE process on the GPU box's host cores).
    python tools/install_cpu_k22.py <bench line file> [<out>]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r04_cpu_k22.json")
d = json.loads([ln for ln in open(src) if ln.startswith("{")][-1])
cb = d["cpu_baseline"]
assert cb["measured_k"] == 22 and cb["scale"] == 1.0 and cb.get("k20_s"), cb
out = {"what": "bench.py --cpu-baseline-k 22 --no-other-configs on the GPU box (gpurun, round 4): real passes of the CPU oracle (oracle/zkoracle.c, OpenMP) "
               "over the headline configuration at k = 18 (median of 3), k = 20 and k = 22 (one pass each) in one process; SRS / keygen excluded; no extrapolation",
       "cpu_baseline": cb, "gpu_value_same_run_s": d["value"], "n_gpus": d["n_gpus"], "config": d["config"]["workload"], "build": d["build"],
       "ratios": {"k22_over_k20": round(cb["value"] / cb["k20_s"], 4), "k20_over_k18": round(cb["k20_s"] / cb["k18_s"], 4)}}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out["ratios"]), cb["value"], cb["k20_s"], cb["k18_s"], cb["cores"])
