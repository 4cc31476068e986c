#!/usr/bin/env python3
"""Reads a rocprofv3 kernel-trace CSV and reports, for the last proof pass in it, the GPU busy time (union of kernel
intervals), the idle gaps and the kernels around the largest gaps.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
    python tools/trace_gaps.py gpurun_out/trace/**/*_kernel_trace.csv"""
import csv
import sys

rows = []
for path in sys.argv[1:]:
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r.get("Stream_Id", "")))
rows.sort()
# a pass starts at each k_sort_hi<false> preceded by > 5 launches of other kernels ... simpler: split at the advice commit = every 6th MSM
starts = [i for i, r in enumerate(rows) if r[2].startswith("void k_sort_hi<false>") or r[2].startswith("k_sort_hi<false>")]
per_pass = 6
if len(starts) >= 2 * per_pass:
    lo = starts[-per_pass]
    # include the NTTs issued before the first MSM of the pass: walk back to the previous k_final_sum
    j = lo
    while j > 0 and "k_final_sum" not in rows[j - 1][2]:
        j -= 1
    rows = rows[j:]
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
gaps = []
prev = rows[0][2]
for s, e, name, st in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, cur_e - t0, name, prev))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    prev = name
busy += cur_e - cur_s
print(f"span {(t1 - t0) / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, idle {(t1 - t0 - busy) / 1e6:.3f} ms in {len(gaps)} gaps, {len(rows)} kernels")
for g, at, name, prev in sorted(gaps, reverse=True)[:25]:
    print(f"  gap {g / 1e3:8.1f} us at +{at / 1e6:7.3f} ms after {prev[:28]:28s} before {name[:40]}")
tot = {}
for s, e, name, st in rows:
    tot[name] = tot.get(name, 0) + e - s
for name, v in sorted(tot.items(), key=lambda kv: -kv[1])[:30]:
    print(f"  {v / 1e6:8.3f} ms  {name[:80]}")
