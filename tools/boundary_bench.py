#!/usr/bin/env python3
"""The host-pointer entry points of the drop-in boundary, timed on their own: zkhip_msm_g1 and zkhip_fft with the caller's
array in PAGEABLE host memory (a Rust Vec<Fr>: what integration/rust/halo2curves-zkhip/src/zkhip.rs hands over, reached from
/root/reference/src/helpers.rs:233,299 and src/bin/cli.rs:320,369,519 through best_multiexp / best_fft), in pinned host memory, and the
device-resident one-column forms of the same work beside them.  Median of --reps calls after a warm-up, wall clock around the blocking call.
    python tools/boundary_bench.py [--k 17 22] [--reps 9] [--chunks 1 2 4 8]     (chunks: the library's msm_host_chunks option — pieces of zkhip_msm_g1's upload + MSM pipeline; 1 = unpipelined)
Prints one JSON object."""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]


def med(f, reps):
    f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        ts.append((time.perf_counter() - t0) * 1000.0)
    return round(statistics.median(ts), 3), round(min(ts), 3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, nargs="+", default=[17, 22])
    ap.add_argument("--reps", type=int, default=9)
    ap.add_argument("--chunks", type=int, nargs="+", default=None, help="values of the library's msm_host_chunks option to time (default: the library's own choice by size)")
    args = ap.parse_args()
    import numpy as np
    import torch

    import halo2_zkcert_amd.ffi as ffi
    import zkoracle_py as zo

    ctx = ffi.Context(0)
    out = {"what": "host-pointer entry points (zkhip_msm_g1, zkhip_fft) against the device-resident one-column forms; ms, median (min) of %d calls" % args.reps, "sizes": {}}
    for k in args.k:
        n = 1 << k
        params = ffi.ParamsKZG.setup(ctx, k, zo.fr_from_int(0x5EED0000 + k))
        dom = ffi.EvaluationDomain(ctx, 4, k)
        d_col = ctx.synth_fill(n, 4242 + k)
        pageable = ctx.to_host(d_col).copy()                 # numpy-owned, pageable
        pin_t = torch.empty((n, 4), dtype=torch.int64).pin_memory()
        pinned = pin_t.numpy().view(np.uint64)
        pinned[:] = pageable
        res = {"bytes": n * 32}
        want = ffi.g1_to_affine(ctx.to_host(params.commit_batch_device([d_col]))[0])

        def dev_msm():
            params.commit_batch_device([d_col])
            ctx.synchronize()
        res["msm_device_resident_one_column"] = med(dev_msm, args.reps)
        d_poly = d_col.clone()

        def dev_fft():
            dom.lagrange_to_coeff_device([d_poly])
            ctx.synchronize()
        res["intt_device_resident_one_column"] = med(dev_fft, args.reps)
        for mode in (args.chunks if args.chunks is not None else [None]):
            tag = "" if mode is None else f"_chunks{mode}"
            if mode is not None:
                ctx.set_option("msm_host_chunks", mode)
            for src_name, src in (("pageable", pageable), ("pinned", pinned)):      # parity first; a mismatch is recorded (and retried once) instead of ending the run
                if not (ffi.g1_to_affine(params.commit(src)) == want).all():
                    again = bool((ffi.g1_to_affine(params.commit(src)) == want).all())
                    out.setdefault("MISMATCHES", []).append(dict(k=k, chunks=mode, source=src_name, repeat_ok=again, source_intact=bool((pinned == pageable).all())))
            res["zkhip_msm_g1_pageable" + tag] = med(lambda: params.commit(pageable), args.reps)
            res["zkhip_msm_g1_pinned" + tag] = med(lambda: params.commit(pinned), args.reps)
            # zkhip_lagrange_to_coeff: best_fft + the 1/n scaling on a host array, in place (the reference's EvaluationDomain::lagrange_to_coeff)
            work = pageable.copy()                          # (the host forms transform IN PLACE: never on the arrays the MSM parity is checked with)
            pin_work_t = torch.empty((n, 4), dtype=torch.int64).pin_memory()
            pin_work = pin_work_t.numpy().view(np.uint64)
            pin_work[:] = pageable
            ref = dom.lagrange_to_coeff(pageable)
            dref = [d_col.clone()]
            dom.lagrange_to_coeff_device(dref)
            assert (ctx.to_host(dref[0]) == ref).all(), "zkhip_lagrange_to_coeff (host) differs from the device form"

            def host_fft(buf):
                ffi._check(ffi.lib().zkhip_lagrange_to_coeff(ctx.h, dom.h, ffi._p(buf)))
            res["zkhip_lagrange_to_coeff_pageable" + tag] = med(lambda: host_fft(work), args.reps)
            res["zkhip_lagrange_to_coeff_pinned" + tag] = med(lambda: host_fft(pin_work), args.reps)
        ctx.set_option("msm_host_chunks", 0)
        dev = res["msm_device_resident_one_column"][0]
        res["msm_pageable_over_device_resident"] = {k_[len("zkhip_msm_g1_pageable"):] or "default": round(v_[0] / dev, 3) for k_, v_ in res.items() if k_.startswith("zkhip_msm_g1_pageable")}
        out["sizes"][f"k{k}"] = res
        params.free()
        dom.free()
        del d_col, d_poly, pin_t
        torch.cuda.empty_cache()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
