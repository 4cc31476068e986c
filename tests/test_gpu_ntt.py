"""GPU parity: best_fft and the EvaluationDomain basis changes vs golden vectors and the CPU oracle."""
import hashlib

import numpy as np
import pytest

from util import H, load

pytestmark = pytest.mark.gpu


def _digest(ints):
    h = hashlib.sha256()
    for v in ints:
        h.update(v.to_bytes(32, "little"))
    return h.hexdigest()


def test_fft_golden(zk, oracle):
    ffi, ctx = zk
    zo = oracle
    g = load("ntt.json")
    for c in g["small"]:
        a = zo.fr_arr_from_ints([H(x) for x in c["input"]])
        out = ctx.best_fft(a, zo.root_of_unity(c["k"]), c["k"])
        assert zo.fr_arr_to_ints(out) == [H(x) for x in c["output"]]
    for c in g["seeded"]:
        n = 1 << c["k"]
        a = zo.synth_raw253(c["seed"], n)
        out = zo.fr_arr_to_ints(ctx.best_fft(a, zo.root_of_unity(c["k"]), c["k"]))
        assert out[:4] == [H(x) for x in c["fft_first"]]
        assert _digest(out) == c["fft_sha256"]
        dom = ffi.EvaluationDomain(ctx, 3, c["k"])
        inv = zo.fr_arr_to_ints(dom.lagrange_to_coeff(a))
        assert _digest(inv) == c["ifft_sha256"]
        dom.free()


@pytest.mark.parametrize("log_n", [0, 1, 2, 5, 9, 11, 12, 13, 16, 18])
def test_fft_vs_oracle(zk, oracle, log_n):
    """covers the 1-pass (<= 2^11), 2-pass and 3-pass plans"""
    ffi, ctx = zk
    zo = oracle
    a = zo.synth_raw253(3000 + log_n, 1 << log_n)
    w = zo.root_of_unity(log_n)
    assert (ctx.best_fft(a, w, log_n) == zo.best_fft(a, w, log_n, 8)).all()


@pytest.mark.parametrize("log_n,smax,r8", [(13, "9", "0"), (16, "9", "0"), (18, "6", "1"), (12, "4", "1"), (15, "5", "1"),
                                            (14, "7", "1"), (20, "11", "1"), (21, "11", "1"), (19, "10", "1"),
                                            (12, "4", "2"), (15, "5", "2"), (14, "7", "2"), (20, "11", "2"), (21, "11", "2"), (22, "11", "2"), (18, "6", "2"),
                                            (12, "4", "3"), (15, "5", "3"), (14, "7", "3"), (20, "10", "3"), (21, "7", "3"), (19, "10", "3"), (17, "9", "3"),
                                            (22, "0", "4"), (19, "0", "4"), (17, "0", "4")])
def test_fft_tile_plans(zk, oracle, log_n, smax, r8):
    """every shape of the tile transform: the stage-per-barrier kernels (ntt_r8 = 0) and the register-tiled ones — 8 elements per thread
    (1: stage groups 3+1, 3+2, 3+3, 3+3+1 ... 3+3+3+2), 4 per thread on 2048-element tiles (2: groups of 2, the last of 1 or 2) and on
    1024-element tiles (3), and the default choice between them (4).  ntt_smax changes the digits of the pass plan (zkhip_set_option, the
    knobs are no longer read from the environment per call)"""
    ffi, ctx = zk
    zo = oracle
    ctx.set_option("ntt_smax", int(smax))
    ctx.set_option("ZKHIP_NTT_R8", int(r8))      # the environment spelling is accepted too
    try:
        a = zo.synth_raw253(3200 + log_n, 1 << log_n)
        w = zo.root_of_unity(log_n)
        assert (ctx.best_fft(a, w, log_n) == zo.best_fft(a, w, log_n, 8)).all()
    finally:
        ctx.set_option("ntt_smax", 0)
        ctx.set_option("ntt_r8", 4)


def test_fft_batch_device(zk, oracle):
    ffi, ctx = zk
    zo = oracle
    log_n = 13
    hs = [zo.synth_raw253(3100 + j, 1 << log_n) for j in range(3)]
    ds = [ctx.to_device(h) for h in hs]
    w = zo.root_of_unity(log_n)
    ctx.fft_batch_device(ds, w, log_n)
    for h, d in zip(hs, ds):
        assert (ctx.to_host(d) == zo.best_fft(h, w, log_n, 8)).all()


def test_domain_golden(zk, oracle):
    ffi, ctx = zk
    zo = oracle
    for c in load("domain.json"):
        dom = ffi.EvaluationDomain(ctx, c["j"], c["k"])
        assert dom.extended_k == c["extended_k"]
        coeffs = zo.fr_arr_from_ints([H(x) for x in c["coeffs"]])
        assert zo.fr_arr_to_ints(dom.lagrange_to_coeff(coeffs)) == [H(x) for x in c["lagrange_to_coeff"]]
        assert zo.fr_arr_to_ints(dom.coeff_to_extended(coeffs)) == [H(x) for x in c["coeff_to_extended"]]
        ext = zo.fr_arr_from_ints([H(x) for x in c["extended_in"]])
        assert zo.fr_arr_to_ints(dom.extended_to_coeff(ext)) == [H(x) for x in c["extended_to_coeff"]]
        d = ctx.to_device(ext)
        dom.divide_by_vanishing_poly_device(d)
        assert zo.fr_arr_to_ints(ctx.to_host(d)) == [H(x) for x in c["divide_by_vanishing"]]
        dom.free()


@pytest.mark.parametrize("j,k", [(4, 10), (5, 11), (9, 9), (3, 13)])
def test_domain_vs_oracle(zk, oracle, j, k):
    ffi, ctx = zk
    zo = oracle
    dom = ffi.EvaluationDomain(ctx, j, k)
    odom = zo.Domain(j, k)
    assert dom.extended_k == odom.extended_k and (dom.extended_omega == odom.extended_omega).all()
    coeffs = zo.synth_raw253(4000 + k, 1 << k)
    ext = dom.coeff_to_extended(coeffs)
    assert (ext == odom.coeff_to_extended(coeffs, 8)).all()
    # ragged input: fewer coefficients than n (zero padded), and more than n
    assert (dom.coeff_to_extended(coeffs[:100]) == odom.coeff_to_extended(coeffs[:100], 8)).all()
    big = zo.synth_raw253(4100 + k, (1 << k) + 5)
    assert (dom.coeff_to_extended(big) == odom.coeff_to_extended(big, 8)).all()
    x = zo.synth_raw253(4200 + k, dom.extended_n)
    assert (dom.extended_to_coeff(x, full=True) == odom.extended_to_coeff_full(x, 8)).all()
    assert (dom.lagrange_to_coeff(coeffs) == odom.lagrange_to_coeff(coeffs, 8)).all()
    # extended_to_coeff(coeff_to_extended(p)) == p padded with zeros
    back = dom.extended_to_coeff(ext, full=True)
    assert (back[: 1 << k] == coeffs).all() and (back[1 << k:] == 0).all()
    dom.free()


@pytest.mark.parametrize("k,j", [(17, 4), (19, 5), (22, 4)] + ([(24, 4)] if __import__("os").environ.get("ZKHIP_TEST_HUGE") else []))   # 2^26 extended: 4 passes
def test_full_size_round_trip_and_point_checks(zk, oracle, k, j):
    """BASELINE sizes (2^17 -> 2^19, 2^19 -> 2^21, 2^22 -> 2^24 extended): size-independent properties.
    (1) extended_to_coeff(coeff_to_extended(p)) == p; (2) three extended evaluations equal Horner on
    the host at the coset points; (3) coeff_to_lagrange(lagrange_to_coeff(v)) == v."""
    ffi, ctx = zk
    zo = oracle
    dom = ffi.EvaluationDomain(ctx, j, k)
    n = 1 << k
    p = ctx.synth_fill(n, 0xABC0 + k)
    ext = dom.coeff_to_extended_device([p])[0]
    ph = ctx.to_host(p)
    eh = ctx.to_host(ext)
    for i in (0, 1, 12345, dom.extended_n - 1):
        e = np.zeros(4, dtype=np.uint64)
        e[0] = i
        x = zo._binary("zko_fr_mul", dom.g_coset, zo._binary("zko_fr_pow", dom.extended_omega, e))
        assert (eh[i] == zo.eval_polynomial(ph, x)).all()
    dom.extended_to_coeff_device([ext])
    back = ctx.to_host(ext)
    assert (back[:n] == ph).all() and (back[n:] == 0).all()
    v = ctx.synth_fill(n, 0xDEF0 + k)
    vh = ctx.to_host(v)
    dom.lagrange_to_coeff_device([v])
    dom.coeff_to_lagrange_device([v])
    assert (ctx.to_host(v) == vh).all()
    dom.free()
