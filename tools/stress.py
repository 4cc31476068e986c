#!/usr/bin/env python3
"""Randomised parity stress against the oracle (run on a GPU box): MSMs over random sizes / scalar patterns / window widths,
NTTs of every size 2^1..2^16, lookup permutes and divisions.  Exits non-zero on the first mismatch.
    python tools/stress.py [--seconds 240] [--seed 1]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np

import halo2_zkcert_amd.ffi as ffi
import zkoracle_py as zo

R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001


def scalars(rng, n, kind):
    s = zo.synth_raw253(int(rng.integers(1, 1 << 30)), n)
    if kind == "bool":
        s = zo.fr_arr_from_ints([int(x) for x in rng.integers(0, 2, n)])
    elif kind == "small":
        s = zo.fr_arr_from_ints([int(x) for x in rng.integers(0, 1 << 16, n)])
    elif kind == "same":
        s[:] = s[0]
    elif kind == "sparse":
        mask = rng.random(n) < 0.9
        s[mask] = 0
    elif kind == "neg":
        s = zo.fr_arr_from_ints([(R - int(x)) % R for x in rng.integers(0, 1 << 20, n)])
    return s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--kmin", type=int, default=1, help="smallest SRS size 2^k for the MSM cases")
    ap.add_argument("--kmax", type=int, default=13)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    ctx = ffi.Context(0)
    t_end = time.time() + args.seconds
    counts = dict(msm=0, ntt=0, permute=0, division=0)
    params_cache = {}
    while time.time() < t_end:
        # --- MSM: random SRS size (window width follows from it), random prefix length, random scalar pattern
        k = int(rng.integers(args.kmin, args.kmax + 1))
        if k not in params_cache:
            params_cache[k] = ffi.ParamsKZG.setup(ctx, k, zo.fr_from_int(int(rng.integers(2, 1 << 60))))
        p = params_cache[k]
        n = int(rng.integers(1, (1 << k) + 1))
        kind = ["uniform", "bool", "small", "same", "sparse", "neg"][int(rng.integers(0, 6))]
        s = scalars(rng, n, kind)
        bases = p.read_bases(p.g, 0, n)
        got = ffi.g1_to_affine(p.commit(s))
        exp = zo.g1_to_affine(zo.best_multiexp(s, bases, 8))
        assert (got == exp).all(), ("msm", k, n, kind)
        counts["msm"] += 1
        # --- NTT
        kk = int(rng.integers(1, 17))
        a = zo.synth_raw253(int(rng.integers(1, 1 << 30)), 1 << kk)
        w = zo.root_of_unity(kk)
        assert (ctx.best_fft(a.copy(), w, kk) == zo.best_fft(a.copy(), w, kk, 8)).all(), ("ntt", kk)
        counts["ntt"] += 1
        # --- lookup permute with heavy repetition
        kp = int(rng.integers(3, 13))
        bf = int(rng.integers(1, 6))
        npk = 1 << kp
        if npk > bf + 3:
            u = npk - bf - 1
            distinct = int(rng.integers(1, u + 1))
            pool = zo.synth_raw253(int(rng.integers(1, 1 << 30)), distinct)
            tab = pool[rng.integers(0, distinct, npk)]
            tab[:min(distinct, u)] = pool[:min(distinct, u)]
            present = np.unique(tab[:u], axis=0)
            inp = present[rng.integers(0, len(present), npk)]
            bi, bt = zo.synth_raw253(7, bf + 1), zo.synth_raw253(8, bf + 1)
            ea, es = zo.permute_expression_pair(kp, bf, inp, tab, bi, bt)
            ga, gs = ffi.permute_expression_pair_device(ctx, kp, bf, ctx.to_device(inp), ctx.to_device(tab), ctx.to_device(bi), ctx.to_device(bt))
            assert (ctx.to_host(ga) == ea).all() and (ctx.to_host(gs) == es).all(), ("permute", kp, bf, distinct)
            counts["permute"] += 1
        # --- division by X - r
        nd = int(rng.integers(1, 9000))
        q = zo.synth_raw253(int(rng.integers(1, 1 << 30)), nd)
        r = zo.synth_raw253(int(rng.integers(1, 1 << 30)), 1)
        d = ctx.to_device(q)
        ffi.kate_division_device(ctx, [d], [r])
        assert (ctx.to_host(d) == zo.kate_division(q, r)).all(), ("division", nd)
        counts["division"] += 1
    print("stress ok:", counts)


if __name__ == "__main__":
    main()
