#!/usr/bin/env python3
"""Isolated timings of the three hot-path primitives (no overlap between them), for DESIGN.md's per-kernel numbers.
    python tools/kernel_bench.py [--ks 17,19,22]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import halo2_zkcert_amd.ffi as ffi
import halo2_zkcert_amd.prover as pv


def timeit(ctx, fn, reps=3):
    fn()
    ctx.synchronize()
    best = 1e9
    for _ in range(reps):
        ctx.torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        ctx.torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ks", default="17,19,22")
    ap.add_argument("--ncols", default="1,4")
    ap.add_argument("--skip-msm", action="store_true")
    args = ap.parse_args()
    ctx = ffi.Context(0)
    for k in [int(x) for x in args.ks.split(",")]:
        n = 1 << k
        params = ffi.ParamsKZG.setup(ctx, k, pv.fr_from_int_host(0x1234567))
        c, W = params.window()
        for ncols in ([] if args.skip_msm else [int(x) for x in args.ncols.split(",")]):
            cols = [ctx.synth_fill(n, 100 + j) for j in range(ncols)]
            ms = timeit(ctx, lambda: ctx.to_host(params.commit_batch_device(cols)))
            ctx.profile_enable(True)
            ctx.to_host(params.commit_batch_device(cols))
            prof = {nm: round(ctx.profile_read(nm)[0], 3) for nm in ("msm_digits", "msm_plan", "msm_accum_affine", "msm_accum_jac", "msm_tail")}
            ctx.profile_enable(False)
            mads = ncols * n * W * (8 * 81 + 2 * 45 + 9 * 90)   # XYZZ mixed add: 8M + 2S, 9 reductions
            print(f"MSM k={k} c={c} W={W} ncols={ncols}: {ms:.3f} ms total ({ms / ncols:.3f} ms/col)  {prof}  "
                  f"accum {mads / prof['msm_accum_affine'] / 1e9:.2f} Tmad/s", flush=True)
        params.free()
        dom = ffi.EvaluationDomain(ctx, 4, k)
        for npoly in (1, 8):
            polys = [ctx.synth_fill(n, 200 + j) for j in range(npoly)]
            ms = timeit(ctx, lambda: dom.lagrange_to_coeff_device(polys))
            print(f"iNTT k={k} x{npoly}: {ms:.3f} ms  ({npoly * n * 64 / ms / 1e6:.1f} GB/s algorithmic)", flush=True)
            outs = None

            def ext():
                nonlocal outs
                outs = dom.coeff_to_extended_device(polys)

            ms = timeit(ctx, ext)
            print(f"coeff_to_extended k={k}->{dom.extended_k} x{npoly}: {ms:.3f} ms  ({npoly * dom.extended_n * 64 / ms / 1e6:.1f} GB/s algorithmic)", flush=True)
            del outs
        dom.free()
        del polys


if __name__ == "__main__":
    main()
