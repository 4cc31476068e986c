#!/usr/bin/env python3
"""Stress of zkhip_create_proof_ex's host-input paths (round 6: uploads from a worker thread, the random polynomial first and in chunks): the SHA-shaped k = 19 and the
aggregation-shaped k = 20 / 22 proofs with FRESH pageable advice arrays per proof, with and without host blinding, worker thread on / off, registration on / off — every proof's
bytes equal the device-input proof's (same blinding).  python tools/host_inputs_stress.py --seconds 90"""
import argparse, hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import halo2_zkcert_amd.ffi as ffi
import halo2_zkcert_amd.prover as pv
ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=90.0)
args = ap.parse_args()
ctx = ffi.Context(0)
rng = np.random.default_rng(7)
cases = []
for name, sh, kind in (("sha19", pv.CircuitShape.sha256(19, n_advice=32, n_fixed=12), "poseidon"), ("agg20", pv.CircuitShape.agg(20, 3, 1), "evm"), ("agg22", pv.CircuitShape.agg(22, 3, 1), "evm")):
    gp = pv.Prover(pv.GpuBackend(ctx, ffi), sh, satisfiable=True)
    w = gp.witness(0)
    bf, n = sh.blinding_factors, 1 << sh.k
    L = max(1, len(sh.lookups))
    host = dict(lookup_permuted=ctx.to_host(ctx.synth_fill(2 * L * (bf + 1), 11)).copy(), perm_z=ctx.to_host(ctx.synth_fill(sh.n_perm_sets * bf, 12)).copy(),
                lookup_z=ctx.to_host(ctx.synth_fill(L * bf, 13)).copy(), random_poly=ctx.to_host(ctx.synth_fill(n, 14)).copy())
    ref_plain = bytes(gp.prove_native(w, transcript=kind)["proof"])
    ref_blind = bytes(gp.prove_native(w, transcript=kind, blinding=host)["proof"])
    assert ref_plain != ref_blind
    cases.append((name, gp, w, kind, host, ref_plain, ref_blind))
t_end, made, bad = time.time() + args.seconds, 0, 0
while time.time() < t_end:
    name, gp, w, kind, host, ref_plain, ref_blind = cases[int(rng.integers(0, len(cases)))]
    thr, reg, blind = int(rng.integers(0, 2)), int(rng.integers(0, 4) == 0), int(rng.integers(0, 2))
    ctx.set_option("host_copy_thread", thr); ctx.set_option("host_register", reg)
    w.pop("advice_host_pageable", None)      # fresh pageable arrays every proof
    b = dict(host, random_poly=host["random_poly"].copy()) if blind else None
    got = bytes(gp.prove_native(w, transcript=kind, host_inputs="pageable", blinding=b)["proof"])
    made += 1
    if got != (ref_blind if blind else ref_plain):
        bad += 1
        print("MISMATCH", name, dict(thread=thr, register=reg, blinding=blind), flush=True)
print(f"host inputs stress: {made} proofs, {bad} mismatches")
