#!/usr/bin/env python3
"""Stress of the host-slice MSM entry point (zkhip_msm_g1): pageable and pinned sources, every chunk count, sizes 2^17..2^21, interleaved with device-resident
work on the same context; every result is compared with the device-resident one-column MSM of the same scalars.  (Round 6: tools/boundary_bench.py once saw
zkhip_msm_g1 from PINNED memory at 2^20, unpipelined, return another point than the device-resident MSM; not reproduced in 100 focused calls.)
    python tools/host_msm_stress.py --seconds 60 [--null-stream 0]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np
import torch

import halo2_zkcert_amd.ffi as ffi
import zkoracle_py as zo

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=60.0)
ap.add_argument("--null-stream", type=int, default=1, help="1: the context stays on torch's default (NULL) stream, as in a plain script; 0: on a stream of its own")
ap.add_argument("--kmin", type=int, default=17)
ap.add_argument("--kmax", type=int, default=21)
args = ap.parse_args()
if not args.null_stream:
    torch.cuda.set_stream(torch.cuda.Stream())
ctx = ffi.Context(0)
rng = np.random.default_rng(12345)
sizes = {}
for k in range(args.kmin, args.kmax + 1):
    n = 1 << k
    params = ffi.ParamsKZG.setup(ctx, k, zo.fr_from_int(0x5EED0000 + k))
    d_col = ctx.synth_fill(n, 4242 + k)
    pageable = ctx.to_host(d_col).copy()
    pin_t = torch.empty((n, 4), dtype=torch.int64).pin_memory()
    pinned = pin_t.numpy().view(np.uint64)
    pinned[:] = pageable
    want = ffi.g1_to_affine(ctx.to_host(params.commit_batch_device([d_col]))[0])
    dom = ffi.EvaluationDomain(ctx, 4, k)
    sizes[k] = dict(params=params, d_col=d_col, pageable=pageable, pin_t=pin_t, pinned=pinned, want=want, dom=dom, d_poly=d_col.clone())
t_end = time.time() + args.seconds
calls, bad = 0, []
while time.time() < t_end:
    k = int(rng.integers(args.kmin, args.kmax + 1))
    s = sizes[k]
    chunks = int(rng.choice([0, 1, 2, 3, 4, 8]))
    src = "pinned" if rng.integers(0, 2) else "pageable"
    pre = int(rng.integers(0, 4))      # device-side work issued right before the call: none / an iNTT / a device MSM / both
    if pre & 1:
        s["dom"].lagrange_to_coeff_device([s["d_poly"]])
    if pre & 2:
        s["params"].commit_batch_device([s["d_col"]])
    ctx.set_option("msm_host_chunks", chunks)
    got = ffi.g1_to_affine(s["params"].commit(s[src]))
    calls += 1
    if not (got == s["want"]).all():
        again = ffi.g1_to_affine(s["params"].commit(s[src]))
        same_src = bool((s["pinned"] == s["pageable"]).all())
        dev_again = ffi.g1_to_affine(ctx.to_host(s["params"].commit_batch_device([s["d_col"]]))[0])
        bad.append(dict(call=calls, k=k, chunks=chunks, src=src, pre=pre, repeat_ok=bool((again == s["want"]).all()), source_intact=same_src,
                        device_msm_still_ok=bool((dev_again == s["want"]).all())))
        print("MISMATCH", bad[-1], flush=True)
print(f"host msm stress: {calls} calls, {len(bad)} mismatches, null_stream={args.null_stream}")
