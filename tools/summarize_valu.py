#!/usr/bin/env python3
"""Reduces the SQ counter passes of tools/profile_valu.sh to one CSV per configuration: per kernel, the dispatch count and the
per-dispatch average of each counter, plus derived figures:
  valu_busy      = SQ_ACTIVE_INST_VALU * 4 / (SIMDs * GRBM_GUI_ACTIVE / XCDs)   (fraction of SIMD cycles with a VALU instruction in
                   flight; SQ_ACTIVE_INST_* count quad-cycles, GRBM_GUI_ACTIVE is summed over the 8 XCDs: MI355X_MICROARCH.md)
  lanes_active   = SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU * 64)           (lane occupancy of the VALU instructions issued)
  insts_per_wave = SQ_INSTS_VALU / SQ_WAVES
"""
import csv
import glob
import os
import re
import shutil
import sys
from collections import defaultdict

out = sys.argv[1]
SIMDS, XCDS = 1024, 8
cfgs = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
cnts = defaultdict(lambda: defaultdict(lambda: defaultdict(int)))
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    cfg = os.path.basename(d).split("_")[1]
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
                name = re.sub(r"<.*", "", name)
                cfgs[cfg][name][r["Counter_Name"]] += float(r["Counter_Value"])
                cnts[cfg][name][r["Counter_Name"]] += 1
for cfg, kernels in cfgs.items():
    names = sorted({c for k in kernels.values() for c in k})
    with open(os.path.join(out, f"valu_{cfg}.csv"), "w") as fh:
        fh.write("kernel,dispatches," + ",".join(names) + ",valu_busy,lanes_active,valu_insts_per_wave\n")
        rows = []
        for k, v in kernels.items():
            n = max(cnts[cfg][k].values())
            avg = {c: v[c] / max(1, cnts[cfg][k][c]) for c in names}
            gui = avg.get("GRBM_GUI_ACTIVE", 0) / XCDS
            busy = avg.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (SIMDS * gui) if gui else 0
            lanes = avg.get("SQ_THREAD_CYCLES_VALU", 0) / (avg.get("SQ_ACTIVE_INST_VALU", 0) * 64) if avg.get("SQ_ACTIVE_INST_VALU") else 0
            ipw = avg.get("SQ_INSTS_VALU", 0) / avg.get("SQ_WAVES", 1) if avg.get("SQ_WAVES") else 0
            rows.append((avg.get("SQ_BUSY_CYCLES", 0) * n, k, n, avg, busy, lanes, ipw))
        for _, k, n, avg, busy, lanes, ipw in sorted(rows, reverse=True):
            fh.write(f"{k},{n}," + ",".join(f"{avg[c]:.0f}" for c in names) + f",{busy:.4f},{lanes:.4f},{ipw:.1f}\n")
for d in glob.glob(os.path.join(out, "pmc_*")):
    if os.path.isdir(d):
        shutil.rmtree(d)
