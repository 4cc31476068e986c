#!/usr/bin/env python3
"""Isolated NTT timings per size and pass plan (ntt_smax = most bits per pass): batches of 8 transforms of 2^m.
    python tools/ntt_plan_bench.py [--ms 17,18,...]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import halo2_zkcert_amd.ffi as ffi
import halo2_zkcert_amd.prover as pv

ap = argparse.ArgumentParser()
ap.add_argument("--ms", default="17,18,19,20,21,22,23,24")
ap.add_argument("--smax", default="9,10,11")
a = ap.parse_args()
ctx = ffi.Context(0)
torch = ctx.torch
for m in [int(x) for x in a.ms.split(",")]:
    n = 1 << m
    w = pv.fr_from_int_host(pow(pv.ROOT_OF_UNITY, 1 << (28 - m), pv.R))
    polys = [ctx.synth_fill(n, 50 + j) for j in range(8)]
    row = []
    for smax in [int(x) for x in a.smax.split(",")]:
        ctx.set_option("ntt_smax", smax)
        for _ in range(2):
            ctx.fft_batch_device(polys, w, m)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            ctx.fft_batch_device(polys, w, m)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        row.append(f"smax {smax}: {best * 1e3:8.3f} ms")
    print(f"2^{m} x 8   " + "   ".join(row), flush=True)
