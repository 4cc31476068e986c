#!/usr/bin/env python3
"""reduces tools/sort_write_probe.sh's rocprofv3 directories: per variant and sort kernel the per-dispatch averages of the write-request counters
(L2 -> fabric write requests, of which 64-byte ones), WRITE_SIZE / FETCH_SIZE (KiB) and the kernel's average duration"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

out = sys.argv[1]


def kname(s):
    s = re.sub(r"\(.*", "", s).replace("void ", "").strip()
    return s


for v in ("tile8k", "tile16k"):
    tot, cnt = defaultdict(float), defaultdict(int)
    for d in glob.glob(os.path.join(out, f"*_{v}")):
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                key = (kname(r["Kernel_Name"]), r["Counter_Name"])
                tot[key] += float(r["Counter_Value"])
                cnt[key] += 1
    dur = {}
    for f in glob.glob(os.path.join(out, f"stats_{v}", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            dur[kname(r["Name"])] = (float(r["AverageNs"]) / 1e6, int(r["Calls"]))
    print(f"== {v}")
    for k in sorted({k for k, _ in tot}):
        if "sort" not in k and "accum_affine" not in k:
            continue
        row = {c: tot[(k, c)] / max(cnt[(k, c)], 1) for (kk, c) in tot if kk == k}
        wr, wr64 = row.get("TCC_EA0_WRREQ_sum", 0), row.get("TCC_EA0_WRREQ_64B_sum", 0)
        d = dur.get(k, (0, 0))
        print(f"{k[:70]:70s} avg {d[0]:.3f} ms x{d[1]}: WRREQ {wr:.3e} (64B: {wr64:.3e} = {100 * wr64 / wr if wr else 0:.1f} %), bytes by size = {(wr - wr64) * 32 + wr64 * 64:.3e}, "
              f"WRITE_SIZE {row.get('WRITE_SIZE', 0) * 1024:.3e} B, FETCH_SIZE {row.get('FETCH_SIZE', 0) * 1024:.3e} B")
