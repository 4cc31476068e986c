#!/usr/bin/env python3
"""gpurun_out/<tag>/detail/*.json (tools/rank_replay_r06.sh) -> profiles/<tag>_rank_replay.json: every rank of the sharded proofs replayed alone, composed
two ways per (N, wire point):
    max_rank_sum_ms        max over ranks of the rank's whole step (round 5's figure: every rank runs ahead freely — an UPPER bound on the speed-up)
    synchronised_step_ms   sum over the proof's exchanges (in issue order, the same on every rank) of the max over ranks of the span between "exchanges 0..i-1 have all
                           completed" and "exchanges 0..i have all completed" (a running max: the bulk communicator's transfers overlap the first communicator's,
                           so completions are not monotone in issue order), + the max tail after the last one: every prefix of the exchange sequence a full
                           barrier across the ranks — a LOWER bound on the speed-up
The real step lies between the two.  Spans come from zkhip_comm_trace (a timing event on the communicator's stream behind every exchange), median of
3 untimed passes per rank.  NOT an N-GPU measurement: single-rank replays on one MI355X, peers fabricated, wire modelled.
    python tools/install_rank_replay_r06.py r06"""
import glob
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
WIRES = {"w0": "none (an exchange costs only the fabricating fill)", "w10": "10 us + bytes / 100 GB/s per link", "w20": "20 us + bytes / 50 GB/s per link",
         "w40": "40 us + bytes / 25 GB/s per link"}
runs = {}
for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", tag, "detail", "*.json"))):
    try:
        runs[os.path.basename(f)[:-5]] = json.load(open(f))
    except ValueError:
        pass


def timeline(d):
    """-> (median done_us per exchange, median end_us, exchange labels) of a replay run's trace passes"""
    tr = (d.get("replay") or {}).get("trace") or []
    tr = [p for p in tr if p.get("done_us")]
    if not tr:
        return None
    m = min(len(p["done_us"]) for p in tr)
    done = [statistics.median(p["done_us"][i] for p in tr) for i in range(m)]
    return done, statistics.median(p["end_us"] for p in tr), tr[0]["exchanges"][:m]


def compose(prefix, n, w, single_ms, phase_key=None):
    """the ranks of one (N, wire) row -> dict"""
    per = {}
    for r in range(n):
        d = runs.get(f"{prefix}_rank{r}_{w}")
        if d is not None:
            per[r] = d
    if len(per) < n:
        return None
    if phase_key:      # the chain: value = the aggregation phase on this rank (host clock between the barrier and the end of the step)
        sums = {r: d["phase_ms_per_step"][phase_key] for r, d in per.items()}
    else:
        sums = {r: d["ms_per_step"] for r, d in per.items()}
    tls = {r: timeline(d) for r, d in per.items()}
    row = {"N": n, "wire": WIRES[w], "ranks_replayed": sorted(per), "rank_ms": {str(r): round(v, 3) for r, v in sums.items()},
           "max_rank_sum_ms": round(max(sums.values()), 3)}
    if all(tls.values()):
        m = min(len(t[0]) for t in tls.values())
        labels = tls[0][2][:m]
        def prefix_done(done):      # when exchanges 0..i have ALL completed on this rank
            out_, cur = [], 0.0
            for v in done[:m]:
                cur = max(cur, v)
                out_.append(cur)
            return out_
        pref = {r: prefix_done(t[0]) for r, t in tls.items()}
        spans = {r: [p_[0]] + [p_[i] - p_[i - 1] for i in range(1, m)] for r, p_ in pref.items()}
        tails = {r: t[1] - pref[r][m - 1] for r, t in tls.items()}
        sync_us = sum(max(spans[r][i] for r in spans) for i in range(m)) + max(tails.values())
        # in between: only the FIRST communicator's exchanges are barriers (their results feed the transcript, so the host really waits for them); the bulk communicator's
        # row windows complete in the background and are first read by the sweep (the first exchange of the quotient phase closes them)
        keep = [i for i in range(m) if not labels[i][2]]
        def first_comm_prefix(done):
            out_, cur = [], 0.0
            for i in range(m):
                if labels[i][2] and labels[i][0] != "quotient":
                    continue
                cur = max(cur, done[i], max([done[j] for j in range(i) if labels[j][2]] or [0.0]) if labels[i][0] == "quotient" else max(cur, done[i]))
                out_.append(cur)
            return out_
        pf1 = {r: first_comm_prefix(t[0]) for r, t in tls.items()}
        m1 = min(len(v) for v in pf1.values())
        sync1_us = sum(max((pf1[r][i] - (pf1[r][i - 1] if i else 0.0)) for r in pf1) for i in range(m1)) + max(t[1] - pf1[r][m1 - 1] for r, t in tls.items())
        by_phase = {}
        for i in range(m):
            ph = labels[i][0] or "-"
            by_phase[ph] = by_phase.get(ph, 0.0) + max(spans[r][i] for r in spans)
        by_phase["after the last exchange"] = max(tails.values())
        ends = {r: t[1] for r, t in tls.items()}
        # the traced passes carry the events' cost and no warm cache of the timed loop: scale the composition by (timed step / traced step) of the slowest rank
        slow = max(sums, key=sums.get)
        k_ = sums[slow] / (ends[slow] / 1000.0) if ends[slow] > 0 else 1.0
        row.update({"exchanges": m, "exchange_count_equal_on_all_ranks": len({len(t[0]) for t in tls.values()}) == 1,
                    "traced_end_ms": {str(r): round(v / 1000.0, 3) for r, v in ends.items()},
                    "synchronised_step_ms_traced": round(sync_us / 1000.0, 3), "timed_over_traced": round(k_, 4),
                    "synchronised_step_ms": round(sync_us / 1000.0 * k_, 3),
                    "synchronised_first_communicator_only_ms": round(sync1_us / 1000.0 * k_, 3),
                    "synchronised_by_phase_ms": {p: round(v / 1000.0 * k_, 3) for p, v in by_phase.items()},
                    "slowest_rank_per_exchange": [max(spans, key=lambda r: spans[r][i]) for i in range(m)]})
    if single_ms:
        row["speedup_upper"] = round(single_ms / row["max_rank_sum_ms"], 2)
        if "synchronised_step_ms" in row:
            row["speedup_lower"] = round(single_ms / row["synchronised_step_ms"], 2)
    return row


out = {"what": "SINGLE-RANK REPLAY on one MI355X, every rank of the proof replayed alone (bench.py --replay-rank R --of N; tools/replay_rccl fabricates the peers, "
               "holds the communicator's stream for the modelled wire).  Two compositions per row: max over ranks of whole steps (ranks run ahead freely: the speed-up's "
               "upper bound) and sum over exchanges of the max over ranks of the span between consecutive exchange completions (every exchange a barrier: the lower "
               "bound).  NOT an N-GPU measurement; proof bytes are wrong by construction.",
       "command": f"gpurun -- bash tools/rank_replay_r06.sh {tag}; python tools/install_rank_replay_r06.py {tag}", "build": None, "k22": [], "chain": []}
single = runs.get("single_k22")
single_ms = single["ms_per_step"] if single else None
out["single_gpu_k22_ms"] = single_ms
out["build"] = single.get("build") if single else None
for n in (8, 4, 2):
    for w in WIRES:
        row = compose(f"k22_of{n}", n, w, single_ms)
        if row:
            out["k22"].append(row)
chain1 = runs.get("chain_single")
out["single_gpu_chain_ms"] = chain1["ms_per_step"] if chain1 else None
for w in WIRES:
    per = {r: runs.get(f"chain_of8_rank{r}_{w}") for r in range(8)}
    if any(v is None for v in per.values()):
        continue
    leaf = {r: d["phase_ms_per_step"]["leaf_proofs_until_the_barrier"] for r, d in per.items()}
    agg = compose("chain_of8", 8, w, None, phase_key="aggregation_proof")
    row = {"N": 8, "wire": WIRES[w], "leaf_phase_ms": {str(r): round(v, 3) for r, v in leaf.items()}, "leaf_phase_slowest_ms": round(max(leaf.values()), 3),
           "leaf_groups": per[0].get("leaf_groups"), "aggregation": agg,
           "step_ms_upper_speedup": round(max(leaf.values()) + agg["max_rank_sum_ms"], 3)}
    if "synchronised_step_ms" in agg:
        row["step_ms_lower_speedup"] = round(max(leaf.values()) + agg["synchronised_step_ms"], 3)
    if chain1:
        row["speedup_upper"] = round(chain1["ms_per_step"] / row["step_ms_upper_speedup"], 2)
        if "step_ms_lower_speedup" in row:
            row["speedup_lower"] = round(chain1["ms_per_step"] / row["step_ms_lower_speedup"], 2)
    # one rank per leaf (the default since round 6; ranks 0, 1, 6 replayed): the leaf phase is the slowest leaf ALONE (no exchange in it), the aggregation phase as above
    ng = {r: runs.get(f"chain_nogroups_of8_rank{r}_{w}") for r in (0, 1, 6)}
    if all(ng.values()):
        leaf_ng = max(d_["phase_ms_per_step"]["leaf_proofs_until_the_barrier"] for d_ in ng.values())
        e = {"leaf_phase_ms": {str(r): round(d_["phase_ms_per_step"]["leaf_proofs_until_the_barrier"], 3) for r, d_ in ng.items()}, "leaf_phase_slowest_ms": round(leaf_ng, 3),
             "step_ms_upper_speedup": round(leaf_ng + agg["max_rank_sum_ms"], 3)}
        if "synchronised_step_ms" in agg:
            e["step_ms_lower_speedup"] = round(leaf_ng + agg["synchronised_step_ms"], 3)
        if chain1:
            e["speedup_upper"] = round(chain1["ms_per_step"] / e["step_ms_upper_speedup"], 2)
            if "step_ms_lower_speedup" in e:
                e["speedup_lower"] = round(chain1["ms_per_step"] / e["step_ms_lower_speedup"], 2)
        row["one_rank_per_leaf"] = e
    out["chain"].append(row)
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_rank_replay.json"), "w"), indent=1)
for row in out["k22"]:
    print(f"k22 N={row['N']} wire [{row['wire'][:12]}]: max-rank {row['max_rank_sum_ms']} ms, first-communicator barriers {row.get('synchronised_first_communicator_only_ms')} ms, synchronised {row.get('synchronised_step_ms')} ms -> speed-up {row.get('speedup_lower')} - {row.get('speedup_upper')}")
for row in out["chain"]:
    if "one_rank_per_leaf" in row:
        e = row["one_rank_per_leaf"]
        print(f"chain N=8 one rank per leaf, wire [{row['wire'][:12]}]: {e.get('step_ms_lower_speedup')} - {e['step_ms_upper_speedup']} ms -> speed-up {e.get('speedup_lower')} - {e.get('speedup_upper')}")
    print(f"chain N=8 (SHA leaves over rank pairs) wire [{row['wire'][:12]}]: {row.get('step_ms_lower_speedup')} - {row['step_ms_upper_speedup']} ms -> speed-up {row.get('speedup_lower')} - {row.get('speedup_upper')}")
