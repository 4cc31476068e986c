import os, sys, time
sys.path[:0] = ["/root/repo", "/root/repo/oracle"]
import numpy as np
import halo2_zkcert_amd.ffi as ffi, halo2_zkcert_amd.prover as pv
import zkoracle_py as zo
ctx = ffi.Context(0)
k = 17; n = 1 << k
params = ffi.ParamsKZG.setup(ctx, k, pv.fr_from_int_host(0x1234567))
rng = np.random.default_rng(1)
R = pv.R
def col(kind):
    u = rng.random(n)
    vals = []
    rnd = zo.fr_arr_to_ints(zo.synth_raw253(int(rng.integers(1, 1 << 30)), n))
    for i in range(n):
        if kind == "sha_bits":   # 90 % bits, 10 % words < 2^32
            vals.append(int(rng.integers(0, 2)) if u[i] < 0.9 else int(rng.integers(0, 1 << 32)))
        elif kind == "uniform" or u[i] < 0.7: vals.append(rnd[i] % R)
        elif u[i] < 0.9: vals.append(int(rng.integers(0, 1 << 63)))
        else: vals.append(int(rng.integers(0, 2)))
    return zo.fr_arr_from_ints(vals)
for kind in ("uniform", "survey", "sha_bits"):
    cols_h = [col(kind) for _ in range(4)]
    cols = [ctx.to_device(c) for c in cols_h]
    got = ctx.to_host(params.commit_batch_device(cols))
    bases = params.read_bases(params.g, 0, n)
    exp = zo.g1_to_affine(zo.best_multiexp(cols_h[0], bases, 16))
    assert (ffi.g1_to_affine(got[0]) == exp).all(), kind
    ctx.synchronize()
    best = 1e9
    for _ in range(5):
        ctx.torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.to_host(params.commit_batch_device(cols))
        best = min(best, time.perf_counter() - t0)
    ctx.profile_enable(True)
    ctx.to_host(params.commit_batch_device(cols))
    prof = {nm: round(ctx.profile_read(nm)[0], 3) for nm in ("msm_digits", "msm_plan", "msm_accum_affine", "msm_accum_jac", "msm_tail")}
    ctx.profile_enable(False)
    print(kind, "4-column MSM 2^17: %.3f ms (parity ok)" % (best * 1e3), prof)
