"""One-off: 2^21 / 2^23 / 2^24 transforms through the 8- and 4-per-thread kernels: values at a few domain points against Horner on the host
(oracle), and the inverse transform recovers the coefficients.  gpurun -- python tools/ab/big_ntt_check.py"""
import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "oracle")]
import numpy as np
import halo2_zkcert_amd.ffi as ffi
import halo2_zkcert_amd.prover as pv
import zkoracle_py as zo
ctx = ffi.Context(0)
for k in (21, 23, 24):
    n = 1 << k
    for mode in (1, 2, 4):
        ctx.set_option("ntt_r8", mode)
        dom = ffi.EvaluationDomain(ctx, 3, k)
        a = ctx.synth_fill(n, 9000 + k)
        b = a.clone()
        dom.coeff_to_lagrange_device([b])
        # point check: evaluation at omega^j equals Horner on the host for a few j
        host = ctx.to_host(a)
        w = zo.root_of_unity(k)
        lag = ctx.to_host(b)
        for j in (0, 1, 12345 % n, n - 1):
            x = zo.fr_from_int(pow(zo.fr_to_int(w), j, pv.R))
            assert (lag[j] == zo.eval_polynomial(host, x)).all(), (k, mode, j)
        dom.lagrange_to_coeff_device([b])
        assert (ctx.to_host(b) == host).all(), (k, mode)
        dom.free()
        print("ok", k, mode, flush=True)
