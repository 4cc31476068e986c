#!/usr/bin/env python3
"""Copies one generation of tools/profile_round.sh output into profiles/ and prints the figures table for DESIGN.md:
    python tools/install_profiles.py r02_v7 [--remove r02_v6]"""
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag)
for f in ("bench.json", "kernel_bench.txt", "bench_detail.json"):      # bench.json: the stdout line (< 4 KB since round 6); bench_detail.json: the whole result object
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(root, "profiles", f"{tag}_{f}"))
for c in ("rsa17", "sha19", "agg22"):
    shutil.copy(os.path.join(src, f"pmc_{c}.csv"), os.path.join(root, "profiles", f"{tag}_pmc_{c}.csv"))
    shutil.copy(os.path.join(src, f"valu_{c}.csv"), os.path.join(root, "profiles", f"{tag}_valu_{c}.csv"))
    shutil.copy(os.path.join(src, f"{c}_kernel_stats.csv"), os.path.join(root, "profiles", f"{tag}_{c}_kernel_stats.csv"))
if "--remove" in sys.argv:
    old = sys.argv[sys.argv.index("--remove") + 1]
    for f in glob.glob(os.path.join(root, "profiles", f"{old}_*")):
        os.remove(f)
j = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
print("build", j["build"], "value", j["value"], "line bytes", len(json.dumps(j)))
if os.path.exists(os.path.join(src, "bench_detail.json")):
    j = json.load(open(os.path.join(src, "bench_detail.json")))
if "configs" not in j:
    raise SystemExit("no bench_detail.json beside the line: the per-configuration table needs the detail object (python bench.py --detail-out ...)")
for k in ("rsa17", "sha19", "agg22"):
    c = j["configs"][k]
    r = c["rooflines"]
    m, n, s = r["msm_accum_affine"], r["ntt"], r["sweep"]
    ir = m["int_roofline"]
    print(f"| {k} | {c['transcript']} | {c['ms_per_step']:.2f} ms | {c['with_h2d']['value'] * 1e3:.2f} ms | {c['setup_s']:.3f} s / {c['resident_bytes'] / 2**30:.1f} GiB | "
          f"{m['achieved']:.0f} GB/s ({m['frac']:.4f}); traffic {m['traffic'] / m['algorithmic_bytes_per_launch']:.1f}× algorithmic; {ir['achieved']:.1f} Tmad/s ({ir['frac']:.2f}) | "
          f"{n['achieved']:.0f} GB/s ({n['frac']:.3f}); traffic {n['traffic'] / n['algorithmic_bytes_per_launch']:.2f}× | {s['achieved']:.0f} GB/s ({s['frac']:.3f}); traffic {s['traffic'] / s['algorithmic_bytes_per_launch']:.2f}× |")
print("survey", j["configs"]["agg22_survey_witness"]["ms_per_step"], "cpu", j["cpu_baseline"]["value"], j["cpu_baseline"]["measured_s"], j["configs"]["rsa17"]["cpu_baseline"]["value"])
for cfg in ("agg22", "rsa17", "sha19"):
    rows = list(csv.DictReader(open(os.path.join(src, f"{cfg}_kernel_stats.csv"))))
    live = json.loads(open(os.path.join(src, f"stats_{cfg}.json")).read().strip().splitlines()[-1])["roofline"]["avg_launch_ms"]
    for r_ in rows:
        if "k_accum_affine" in r_["Name"]:
            print(cfg, "k_accum_affine rocprof", r_["Calls"], "calls avg", round(float(r_["AverageNs"]) / 1e6, 4), "ms; live", live)
    for r_ in csv.DictReader(l for l in open(os.path.join(src, f"valu_{cfg}.csv")) if not l.startswith("#")):
        if r_["kernel"] in ("k_accum_affine", "k_ntt_strided_r8", "k_ntt_strided_r4", "k_ntt_strided_r4s", "k_ntt_final_r4", "k_ntt_final_r4s", "k_sweep", "k_sort_hi", "k_sort_lo_staged16"):
            print("   ", r_["kernel"], "valu_busy", r_["valu_busy"], "lanes", r_["lanes_active"], "valu/wave", r_["valu_insts_per_wave"])
for k, v in j["configs"]["agg22"]["kernels_ms_per_step"].items():
    print("   ", k, v["ms_per_step"])
