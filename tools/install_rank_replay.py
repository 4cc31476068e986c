#!/usr/bin/env python3
"""gpurun_out/<tag>/rank_replay.jsonl (tools/rank_replay.sh) -> profiles/<tag>_rank_replay.json: per run the rank's time per step, its kernels, what
the RCCL branch exchanged, and the derived table DESIGN.md 7 quotes (what does not divide by N).
    python tools/install_rank_replay.py r05 [more tags ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tags = sys.argv[1:] or ["r05"]      # several: later tags override earlier ones label by label (re-runs of part of the sweep); the first names the output
tag = tags[0]
rows = {}
for t_ in tags:
    for ln in open(os.path.join(ROOT, "gpurun_out", t_, "rank_replay.jsonl")):
        try:
            r_ = json.loads(ln)
        except ValueError:      # a run that printed no line
            continue
        rows[r_["label"]] = dict(r_, tag=t_)
rows = list(rows.values())
out = {"what": "SINGLE-RANK REPLAY on one MI355X (bench.py --replay-rank R --of N; tools/replay_rccl fabricates the peers on the device): what rank R of an "
               "N-rank proof costs alone — its kernels, launch structure and exchanges through csrc/comm.hip's RCCL branch.  NOT an N-GPU measurement; "
               "proof bytes are wrong by construction.  `wire` rows hold the communicator's stream for 20 us + max-over-peers(bytes) / 50 GB/s per exchange.",
       "command": "gpurun -- bash tools/rank_replay.sh <tag>; gpurun -- bash tools/rank_replay_chain.sh <tag>  (`box`: which call a row came from — compare rows of one box)", "runs": {}}
for r in rows:
    d = r["line"]
    cfgs = d.get("configs") or {}
    c = cfgs[next(iter(cfgs))] if cfgs else {}
    e = {"ms_per_step": d["ms_per_step"], "build": d.get("build"), "box": r["tag"]}
    if c.get("kernels_ms_per_step"):
        e["kernels_ms_per_step"] = {k: round(v["ms_per_step"], 3) for k, v in c["kernels_ms_per_step"].items() if v["ms_per_step"] > 0}
        e["native_call_ms_per_step"] = c.get("native_call_ms_per_step")
    if d.get("phase_ms_per_step"):
        e["phase_ms_per_step"] = {k: v for k, v in d["phase_ms_per_step"].items() if k != "note"}
    if d.get("replay"):
        e["replay"] = {"rank": d["replay"]["rank"], "of": d["replay"]["of"], "exchanges_per_step": d["replay"]["exchanges_per_step"],
                       "modelled_wire": {k: v for k, v in d["replay"]["modelled_wire"].items() if k != "note"}}
        cm = d.get("comm") or {}
        e["comm"] = {k: cm.get(k) for k in ("transport", "nranks", "collectives_per_step", "bytes_gathered_per_step", "shard_mode", "exchange_modes")}
    out["runs"][r["label"]] = e
R = out["runs"]
if "single_k22" in R:
    one = R["single_k22"]["ms_per_step"]
    tab = []
    for n in (2, 4, 8):
        a, w = R.get(f"k22_rank0_of{n}"), R.get(f"k22_rank0_of{n}_wire20us_50GBs")
        if a:
            tab.append({"N": n, "rank0_ms": a["ms_per_step"], "ideal_ms": round(one / n, 2), "compute_only_speedup": round(one / a["ms_per_step"], 2),
                        "not_dividing_ms": round(a["ms_per_step"] - one / n, 2), "with_modelled_wire_ms": w and w["ms_per_step"],
                        "speedup_with_modelled_wire": w and round(one / w["ms_per_step"], 2),
                        "received_MB_per_proof": round(a["replay"]["exchanges_per_step"]["bytes_received"] / 1e6, 1),
                        "sent_MB_per_proof": round(a["replay"]["exchanges_per_step"]["bytes_sent"] / 1e6, 1), "exchanges_per_proof": a["replay"]["exchanges_per_step"]["collectives"]})
    out["k22_table"] = {"single_gpu_ms": one, "rows": tab,
                        "note": "rank 0 against ranks 3 and 7 of 8: see the runs (with owner = column mod N rank 0 owned a column of every batch and was 6 ms behind rank 7: the "
                                "`before_round_robin_owners/` rows; dealt round robin across batches the ranks are within 2 ms of each other); compute-only = exchanges cost "
                                "only the fabricating fill, i.e. the ceiling a perfect interconnect would allow"}
ch = {k_: R[k_] for k_ in R if k_.startswith("chain_")}
if "chain_single" in ch:
    def ph(label, key):
        return (ch.get(label) or {}).get("phase_ms_per_step", {}).get(key)
    rows8 = [lab for lab in ch if lab.endswith("_of8")]
    if rows8:
        leaf = max(ph(lab, "leaf_proofs_until_the_barrier") for lab in rows8)
        agg = max(ph(lab, "aggregation_proof") for lab in rows8)
        out["chain_table"] = {"single_gpu_ms": ch["chain_single"]["ms_per_step"],
                              "N8_leaf_phase_ms_slowest_rank": leaf, "N8_aggregation_phase_ms_slowest_rank": agg, "N8_step_ms": round(leaf + agg, 2),
                              "N8_compute_only_speedup": round(ch["chain_single"]["ms_per_step"] / (leaf + agg), 2),
                              "N8_without_leaf_groups_step_ms": ch.get("chain_rank1_of8_no_groups", {}).get("ms_per_step"),
                              "note": "a step = the slowest rank's leaf phase (a barrier ends it) + the slowest rank's share of the aggregation proof; ranks replayed: "
                                      "0 (an RSA leaf), 1 (head of a SHA leaf's pair), 4 (second member of that pair), 6 (no leaf)"}
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_rank_replay.json"), "w"), indent=1)
print(json.dumps(out.get("k22_table"), indent=1))
