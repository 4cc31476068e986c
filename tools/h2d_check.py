#!/usr/bin/env python3
"""with_h2d at k = 22, phase by phase: ZKHIP_HOST_TIMING=1 python tools/h2d_check.py  (prints the host phase clock of resident and host-input proofs)"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import halo2_zkcert_amd.ffi as ffi
import halo2_zkcert_amd.prover as pv

ns = argparse.Namespace(agg_k=22, agg_advice=3, agg_lookup_advice=1, sha_advice=32, sha_fixed=12)
ctx = ffi.Context(0)
p = pv.Prover(pv.GpuBackend(ctx, ffi), bench.make_shape(pv, "agg22", ns), satisfiable=True)
w = p.witness(0)
for host in (False, True):
    for _ in range(2):
        p.prove_native(w, transcript="evm", host_inputs=host)
    ctx.synchronize()
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        sys.stderr.write(f"---- host_inputs={host}\n")
        p.prove_native(w, transcript="evm", host_inputs=host)
        ts.append((time.perf_counter() - t0) * 1e3)
    print("host_inputs", host, [round(t, 2) for t in ts], flush=True)
