#!/usr/bin/env python3
"""Where does `setup_s` of a configuration go?  (a reference command is one proof per process.)
    python tools/setup_breakdown.py [agg22|rsa17|sha19]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import halo2_zkcert_amd.ffi as ffi
import halo2_zkcert_amd.prover as pv

name = sys.argv[1] if len(sys.argv) > 1 else "agg22"
shape = {"agg22": lambda: pv.CircuitShape.agg(22, 3, 1), "rsa17": lambda: pv.CircuitShape.rsa(17), "sha19": lambda: pv.CircuitShape.sha256(19, n_advice=32, n_fixed=12)}[name]()
kind = {"agg22": "evm", "rsa17": "poseidon", "sha19": "poseidon"}[name]


def clock(label, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    print(f"  {label:42s} {time.perf_counter() - t0:8.3f} s", flush=True)
    return r


t_all = time.perf_counter()
ctx = clock("zkhip_init", lambda: ffi.Context(0))
be = pv.GpuBackend(ctx, ffi)
params = clock("ParamsKZG.setup (SRS + window tables)", lambda: ffi.ParamsKZG.setup(ctx, shape.k, be.fr(0x1D5C0FFEE)))
params.free()
ctx.profile_enable(True) if os.environ.get("SETUP_KERNELS") else None
prover = clock("Prover (params again, keygen-shaped setup)", lambda: pv.Prover(be, shape, satisfiable=True))
wit = clock("witness", lambda: prover.witness(0))
clock("first proof", lambda: prover.prove_native(wit, transcript=kind))
clock("second proof", lambda: prover.prove_native(wit, transcript=kind))
print(f"  total {time.perf_counter() - t_all:.3f} s")
