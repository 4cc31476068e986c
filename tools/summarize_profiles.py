#!/usr/bin/env python3
"""Reduces the rocprofv3 output directories of tools/profile_round.sh to small CSVs (the raw traces are tens of MiB):
  <cfg>_kernel_stats.csv   the --stats table of the kernel trace
  pmc_<cfg>.csv            per kernel: dispatches, total and per-dispatch average of FETCH_SIZE / WRITE_SIZE (KiB) / SQ_LDS_BANK_CONFLICT,
                           first line `# build=<hash of the kernel sources>`
  valu_<cfg>.csv           per kernel: the SQ counters and derived VALU figures:
      valu_busy      = SQ_ACTIVE_INST_VALU * 4 / (SIMDs * GRBM_GUI_ACTIVE / XCDs)   (share of SIMD cycles with a VALU instruction in flight;
                       SQ_ACTIVE_INST_* count quad-cycles, GRBM_GUI_ACTIVE is the sum over the 8 XCDs: MI355X_MICROARCH.md)
      lanes_active   = SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU * 64)
      valu_insts_per_wave = SQ_INSTS_VALU / SQ_WAVES"""
import csv
import glob
import os
import re
import shutil
import sys
from collections import defaultdict

out = sys.argv[1]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    from bench import build_hash
    BH = build_hash()
except Exception:   # noqa: BLE001
    BH = "unknown"
SIMDS, XCDS = 1024, 8


def kname(s):
    s = re.sub(r"\(.*", "", s).replace("void ", "").strip()
    return re.sub(r"<.*", "", s)


def collect(dirs):
    tot, cnt = defaultdict(float), defaultdict(int)
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    key = (r["Counter_Name"], kname(r["Kernel_Name"]))
                    tot[key] += float(r["Counter_Value"])
                    cnt[key] += 1
    return tot, cnt


for cfg in ("rsa17", "sha19", "agg22", "k17", "k22"):
    stats = sorted(glob.glob(os.path.join(out, f"stats_{cfg}", "**", "*kernel_stats.csv"), recursive=True))
    if stats:
        shutil.copy(stats[0], os.path.join(out, f"{cfg}_kernel_stats.csv"))
    pd = [d for d in sorted(glob.glob(os.path.join(out, f"pmc_{cfg}_*"))) if os.path.isdir(d)]
    if pd:
        tot, cnt = collect(pd)
        rows = sorted(((c, k, cnt[(c, k)], v) for (c, k), v in tot.items()), key=lambda r: (r[0], -r[3]))
        with open(os.path.join(out, f"pmc_{cfg}.csv"), "w") as fh:
            fh.write(f"# build={BH}\n")
            fh.write("counter,kernel,dispatches,total,avg_per_dispatch\n")
            for c, k, n, v in rows:
                if v > 0:
                    fh.write(f"{c},{k},{n},{v:.0f},{v / n:.1f}\n")
    vd = [d for d in sorted(glob.glob(os.path.join(out, f"valu_{cfg}_*"))) if os.path.isdir(d)]
    if vd:
        tot, cnt = collect(vd)
        kernels = sorted({k for _, k in tot})
        names = sorted({c for c, _ in tot})
        rows = []
        for k in kernels:
            n = max(cnt[(c, k)] for c in names if (c, k) in cnt)
            avg = {c: (tot[(c, k)] / cnt[(c, k)] if cnt.get((c, k)) else 0.0) for c in names}
            gui = avg.get("GRBM_GUI_ACTIVE", 0) / XCDS
            act = avg.get("SQ_ACTIVE_INST_VALU", 0)
            busy = act * 4 / (SIMDS * gui) if gui else 0
            lanes = avg.get("SQ_THREAD_CYCLES_VALU", 0) / (act * 64) if act else 0
            ipw = avg.get("SQ_INSTS_VALU", 0) / avg["SQ_WAVES"] if avg.get("SQ_WAVES") else 0
            rows.append((avg.get("SQ_BUSY_CYCLES", 0) * n, k, n, avg, busy, lanes, ipw))
        with open(os.path.join(out, f"valu_{cfg}.csv"), "w") as fh:
            fh.write(f"# build={BH}\n")
            fh.write("kernel,dispatches," + ",".join(names) + ",valu_busy,lanes_active,valu_insts_per_wave\n")
            for _, k, n, avg, busy, lanes, ipw in sorted(rows, reverse=True):
                fh.write(f"{k},{n}," + ",".join(f"{avg[c]:.0f}" for c in names) + f",{busy:.4f},{lanes:.4f},{ipw:.1f}\n")
for d in glob.glob(os.path.join(out, "stats_*")) + glob.glob(os.path.join(out, "pmc_*")) + glob.glob(os.path.join(out, "valu_*")):
    if os.path.isdir(d):
        shutil.rmtree(d)
