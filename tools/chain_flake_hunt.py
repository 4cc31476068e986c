#!/usr/bin/env python3
"""Hunts the round-6 anomaly: ONE `bench.py --gpus 6 --chain --agg-k 18 --no-ladder` run (six ranks on one device, one rank per leaf) produced a SHA-shaped k = 19 leaf proof whose digest
differed from the single-GPU chain's.  Runs that command --iterations times (alternating with the --leaf-groups form) and compares every leaf digest and the aggregation proof's with the
single-GPU chain's.
    python tools/chain_flake_hunt.py --iterations 30"""
import argparse
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--iterations", type=int, default=30)
ap.add_argument("--ranks", type=int, default=6)
args = ap.parse_args()


def run(extra, env=None):
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "d.json")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--chain", "--steps", "1", "--warmup", "1", "--agg-k", "18", "--no-cpu-baseline", "--detail-out", path] + extra,
                           capture_output=True, text=True, env=env, timeout=900)
        if r.returncode != 0:
            return None, r.stderr[-1500:]
        return json.load(open(path)), ""


one, err = run([])
assert one, err
want_leaves, want_agg = one["proof_sha256"][:4], one["proof_sha256"][4]
env = dict(os.environ, ZKHIP_BENCH_ONE_DEVICE="1", ZKHIP_BENCH_DIST_BACKEND="gloo")
env.pop("WORLD_SIZE", None)
bad = []
for it in range(args.iterations):
    extra = ["--gpus", str(args.ranks), "--no-ladder"] + (["--leaf-groups"] if it % 3 == 2 else [])
    d, err = run(extra, env)
    if d is None:
        bad.append(dict(iteration=it, error=err))
        print("RUN FAILED", it, err[-400:], flush=True)
        continue
    got = [d["leaf_proof_sha256"][str(j)] for j in range(4)]
    ok = got == want_leaves and d["proof_sha256"][-1] == want_agg
    if not ok:
        bad.append(dict(iteration=it, grouped=it % 3 == 2, leaves=[g == w for g, w in zip(got, want_leaves)], agg=d["proof_sha256"][-1] == want_agg, got=got))
        print("MISMATCH", bad[-1], flush=True)
print(json.dumps({"iterations": args.iterations, "ranks": args.ranks, "mismatches": len(bad), "bad": bad}))
