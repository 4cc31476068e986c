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
from verify_util import verify_trace


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
    while time.time() < t_end:
        k = args.kmin + seed % (args.kmax - args.kmin + 1)
        if k not in provers:
            provers[k] = pv.Prover(pv.GpuBackend(ctx, ffi), pv.CircuitShape.small(k), satisfiable=True)
        p = provers[k]
        w = p.witness(seed)
        t = p.prove_native(w)
        assert verify_trace(p, w, t), ("verify", k, seed)
        if seed % 5 == 0:
            assert p.prove(w)["commitments"] == t["commitments"], ("schedule mismatch", k, seed)
        done += 1
        seed += 1
    print("proof stress ok:", done, "proofs verified")


if __name__ == "__main__":
    main()
