#!/usr/bin/env python3
"""Reduces the rocprofv3 output directories of tools/profile_round.sh to the two small CSVs kept under profiles/:
kernel_stats.csv (the --stats table) and pmc_fetch_write.csv (per-kernel totals of each counter pass)."""
import csv
import glob
import os
import re
import shutil
import sys
from collections import defaultdict

out = sys.argv[1]
stats = sorted(glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True))
if stats:
    shutil.copy(stats[0], os.path.join(out, "kernel_stats.csv"))
rows = []
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        tot = defaultdict(float)
        cnt = defaultdict(int)
        with open(f) as fh:
            for r in csv.DictReader(fh):
                name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
                name = re.sub(r"<.*", "", name)
                key = (r["Counter_Name"], name)
                tot[key] += float(r["Counter_Value"])
                cnt[key] += 1
        for (c, k), v in tot.items():
            rows.append((c, k, cnt[(c, k)], v))
rows.sort(key=lambda r: (r[0], -r[3]))
with open(os.path.join(out, "pmc_fetch_write.csv"), "w") as fh:
    fh.write("counter,kernel,dispatches,total,avg_per_dispatch\n")
    for c, k, n, v in rows:
        if v > 0:
            fh.write(f"{c},{k},{n},{v:.0f},{v / n:.1f}\n")
for d in glob.glob(os.path.join(out, "stats")) + glob.glob(os.path.join(out, "pmc_*")):
    if os.path.isdir(d):
        shutil.rmtree(d)   # the raw traces are tens of MiB
