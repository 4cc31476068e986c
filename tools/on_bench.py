#!/usr/bin/env python3
"""Isolated timings of the streaming O(n) kernels of the SHPLONK stage (linear combination, division by X - r, evaluation) with
their byte rates, through the C ABI's device entry points.
    python tools/on_bench.py [--k 22]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import halo2_zkcert_amd.ffi as ffi
import halo2_zkcert_amd.prover as pv


def timeit(ctx, fn, reps=5):
    fn()
    ctx.synchronize()
    best = 1e9
    for _ in range(reps):
        ctx.torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        ctx.synchronize()
        ctx.torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, default=22)
    args = ap.parse_args()
    ctx = ffi.Context(0)
    n = 1 << args.k
    polys = [ctx.synth_fill(n, 300 + j) for j in range(32)]
    sc = np.stack([pv.fr_from_int_host(0x9E3779B97F4A7C15 * (j + 3) + 12345) for j in range(48)])
    for m in (1, 2, 4, 8, 16, 32):
        ms = timeit(ctx, lambda: ffi.linear_combination_device(ctx, polys[:m], sc[:m]))
        print(f"lincomb k={args.k} x{m}: {ms:.3f} ms  {(m + 1) * n * 32 / ms / 1e6:.0f} GB/s", flush=True)
    for m in (1, 4, 8, 16):
        work = [p.clone() for p in polys[:m]]
        ms = timeit(ctx, lambda: ffi.kate_division_device(ctx, work, [[sc[j]] for j in range(m)]))
        print(f"kate k={args.k} x{m} (one root each): {ms:.3f} ms  {2 * m * n * 32 / ms / 1e6:.0f} GB/s (one read + one write)", flush=True)
        del work
    for m in (1, 8, 32):
        ms = timeit(ctx, lambda: ffi.eval_polynomials_at_device(ctx, polys[:m], sc[:m]))
        print(f"eval k={args.k} x{m}: {ms:.3f} ms  {m * n * 32 / ms / 1e6:.0f} GB/s", flush=True)


if __name__ == "__main__":
    main()
