#!/usr/bin/env python3
"""N back-to-back proofs of one bench configuration and nothing else (for kernel traces / gap analysis):
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 tools/trace_one.py --config rsa17 --steps 5
    python tools/trace_gaps.py gpurun_out/trace/**/*_kernel_trace.csv"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import halo2_zkcert_amd.ffi as ffi
import halo2_zkcert_amd.prover as pv

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="rsa17")
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--transcript", default=None)
ap.add_argument("--set", action="append", default=[], help="name=value tuning knob (zkhip_set_option)")
a = ap.parse_args()
ns = argparse.Namespace(agg_k=22, agg_advice=3, agg_lookup_advice=1, sha_advice=32, sha_fixed=12)
ctx = ffi.Context(0)
for kv in a.set:
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
shape = bench.make_shape(pv, a.config, ns)
p = pv.Prover(pv.GpuBackend(ctx, ffi), shape, satisfiable=True)
w = p.witness(0)
kind = a.transcript or bench.TRANSCRIPT[a.config]
for _ in range(3):
    p.prove_native(w, transcript=kind)
ctx.synchronize()
ts = []
for _ in range(a.steps):
    t0 = time.perf_counter()
    p.prove_native(w, transcript=kind)
    ts.append((time.perf_counter() - t0) * 1e3)
print(f"{a.config} {kind}: ms per proof min {min(ts):.3f} median {sorted(ts)[len(ts) // 2]:.3f}", [round(t, 3) for t in ts])
