#!/usr/bin/env python3
"""End-to-end stress (run on a GPU box): many satisfiable instances of varying size / seed through zkhip_create_proof, each proof
checked by the oracle's verifier equations (tests/verify_util.py) and against the step-by-step schedule.
    python tools/stress_proofs.py [--seconds 300] [--kmin 6] [--kmax 11]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import halo2_zkcert_amd.ffi as ffi
import halo2_zkcert_amd.prover as pv
from verify_util import verify_proof, verify_trace


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300)
    ap.add_argument("--kmin", type=int, default=6)
    ap.add_argument("--kmax", type=int, default=11)
    args = ap.parse_args()
    ctx = ffi.Context(0)
    t_end = time.time() + args.seconds
    done, seed = 0, 0
    provers = {}
    kinds = ("poseidon", "evm", "blake2b")
    shapes = {"small": pv.CircuitShape.small, "sha": lambda k: pv.CircuitShape.sha256(k, n_advice=12, n_fixed=5), "agg": pv.CircuitShape.agg,
              "phase": pv.CircuitShape.two_phase}      # an advice column of the second phase + a user challenge
    while time.time() < t_end:
        k = args.kmin + seed % (args.kmax - args.kmin + 1)
        name = ("small", "sha", "agg", "phase")[(seed // 3) % 4]
        if (name, k) not in provers:
            provers[(name, k)] = pv.Prover(pv.GpuBackend(ctx, ffi), shapes[name](k), satisfiable=True)
        p = provers[(name, k)]
        w = p.witness(seed, dist="survey" if name == "agg" and seed % 2 else "uniform")
        kind = kinds[seed % 3]
        t = p.prove_native(w, transcript=kind, host_inputs=seed % 4 == 0)
        assert verify_proof(p, w, t["proof"], kind), ("verify bytes", name, k, seed, kind)
        if seed % 5 == 0:
            assert p.prove(w, transcript=kind)["proof"] == t["proof"], ("schedule mismatch", name, k, seed)
        if seed % 7 == 0:
            assert verify_trace(p, w, p.prove_native(w)), ("verify trace", name, k, seed)
        done += 1
        seed += 1
    print("proof stress ok:", done, "proofs verified")


if __name__ == "__main__":
    main()
