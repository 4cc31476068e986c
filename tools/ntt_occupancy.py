#!/usr/bin/env python3
"""Occupancy-vs-time curve of the register-tiled NTT kernels (would more tiles per CU — a 32-byte LDS element: five
1024-element tiles instead of four — buy anything?).  ZKHIP_NTT_LDS_PAD=<KiB> adds unused dynamic LDS to every workgroup, so the SAME
kernel runs with 4, 3, 2 or 1 tiles per CU; ZKHIP_NTT_R8 picks the tile shape (3: 1024 elements / 36 KiB / 256 threads, 2: 2048 / 72 KiB / 512).
    for pad in 0 17 44 100; do ZKHIP_NTT_R8=3 ZKHIP_NTT_LDS_PAD=$pad python tools/ntt_occupancy.py; done"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import halo2_zkcert_amd.ffi as ffi

k = int(sys.argv[1]) if len(sys.argv) > 1 else 22
ctx = ffi.Context(0)
n = 1 << k
dom = ffi.EvaluationDomain(ctx, 4, k)
polys = [ctx.synth_fill(n, 200 + j) for j in range(8)]


def best(fn, reps=7):
    fn()
    ctx.synchronize()
    b = 1e9
    for _ in range(reps):
        ctx.torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        ctx.torch.cuda.synchronize()
        b = min(b, time.perf_counter() - t0)
    return b * 1e3


r8, pad = int(os.environ.get("ZKHIP_NTT_R8", "4")), int(os.environ.get("ZKHIP_NTT_LDS_PAD", "0"))
tile_kib = 36 if (r8 == 3 or (r8 == 4 and k < 21)) else 72
tiles = int(160 // (tile_kib + pad))
t_i = best(lambda: dom.lagrange_to_coeff_device(polys))
outs = []


def ext():
    outs[:] = dom.coeff_to_extended_device(polys)


t_e = best(ext)
print(f"ntt_r8={r8} tile {tile_kib} KiB + pad {pad:3d} KiB -> {tiles} tile(s) per CU, {tiles * (4 if tile_kib == 36 else 8)} waves per CU: "
      f"iNTT 2^{k} x 8 {t_i:.3f} ms, coeff_to_extended x 8 {t_e:.3f} ms", flush=True)
