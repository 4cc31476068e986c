"""GPU parity: SHPLONK prover arithmetic (linear combinations, kate division, the whole multi-open) vs golden vectors and the oracle."""
import numpy as np
import pytest

from util import H, load

pytestmark = pytest.mark.gpu
R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001


def _case(zo):
    g = load("shplonk.json")
    F = lambda xs: zo.fr_arr_from_ints([H(x) for x in xs])
    return g, F


def test_primitives_golden(zk, oracle):
    ffi, ctx = zk
    zo = oracle
    g, F = _case(zo)
    polys = [ctx.to_device(F(c)) for c in g["polys"]]
    lc = g["lincomb"]
    out = ffi.linear_combination_device(ctx, polys, F(lc["coeffs"]), F(lc["low"]))
    assert zo.fr_arr_to_ints(ctx.to_host(out)) == [H(x) for x in lc["out"]]
    dv = g["division"]
    a = ctx.to_device(F(dv["dividend"]))
    ffi.kate_division_device(ctx, [a], [F(dv["roots"])])
    assert zo.fr_arr_to_ints(ctx.to_host(a)) == [H(x) for x in dv["quotient"]]


def test_multiopen_golden(zk, oracle):
    import halo2_zkcert_amd.prover as pv
    import halo2_zkcert_amd.shplonk as sp

    ffi, ctx = zk
    zo = oracle
    g, F = _case(zo)
    b = pv.GpuBackend(ctx, ffi)
    b.setup(g["k"], 3, H(g["s"]))
    polys = {i: ctx.to_device(F(c)) for i, c in enumerate(g["polys"])}
    queries = [(i, H(pt), H(e)) for i, pt, e in g["queries"]]
    ch = {"shplonk_y": H(g["y"]), "shplonk_v": H(g["v"]), "shplonk_u": H(g["u"])}
    pr = sp.ProverSHPLONK(b).create_proof(polys, queries, lambda t: ch[t], lambda t, c: None)
    assert zo.fr_arr_to_ints(ctx.to_host(pr["h_x"])) == [H(x) for x in g["h"]]
    assert zo.fr_arr_to_ints(ctx.to_host(pr["l_x"])) == [H(x) for x in g["h_prime"]]
    for got, exp in ((pr["h1"], g["h1"]), (pr["h2"], g["h2"])):
        assert zo.affine_to_ints(np.asarray(got[0]).reshape(1, 8))[0] == (H(exp[0]), H(exp[1]))


@pytest.mark.parametrize("n,npolys,nlow", [(1, 1, 1), (100, 3, 4), (2048, 4, 0), (5000, 7, 2), (1 << 15, 70, 3)])
def test_linear_combination_vs_oracle(zk, oracle, n, npolys, nlow):
    ffi, ctx = zk
    zo = oracle
    polys = [zo.synth_raw253(7000 + j, n) for j in range(npolys)]
    cf = zo.synth_raw253(7100 + n, npolys)
    cf[0] = 0
    low = zo.synth_raw253(7200 + n, nlow) if nlow else None
    exp = zo.linear_combination(polys, cf, low)
    got = ffi.linear_combination_device(ctx, [ctx.to_device(q) for q in polys], cf, low)
    assert (ctx.to_host(got) == exp).all()


@pytest.mark.parametrize("n", [1, 2, 9, 2047, 2048, 2049, 1 << 14, (1 << 17) + 13, 1 << 20])
def test_kate_division_vs_oracle(zk, oracle, n):
    """several polynomials with different numbers of roots in one call; dividends are arbitrary (remainders dropped)"""
    ffi, ctx = zk
    zo = oracle
    polys = [zo.synth_raw253(7300 + j, n) for j in range(4)]
    roots = [zo.synth_raw253(7400 + j, c) for j, c in enumerate((1, 3, 0, 4))]
    roots[1][1] = 0                                  # division by X
    roots[3][2] = zo.fr_from_int(1)                  # and by X - 1
    dev = [ctx.to_device(q) for q in polys]
    ffi.kate_division_device(ctx, dev, roots)
    for q, r, d in zip(polys, roots, dev):
        exp = zo.kate_division(q, r) if len(r) else q
        assert (ctx.to_host(d) == exp).all()


def test_division_is_exact_on_multiples(zk, oracle):
    """k = 17: (X - r1)(X - r2) q(X) divided back gives q(X) — a size-independent property at the benchmark's size"""
    ffi, ctx = zk
    zo = oracle
    n = 1 << 17
    q = zo.synth_raw253(7500, n)
    q[n - 2:] = 0
    r = zo.synth_raw253(7501, 2)
    # multiply by (X - r): shift minus r * q, twice, with the oracle's field ops on whole columns
    cur = q.copy()
    for j in range(2):
        shifted = np.zeros_like(cur)
        shifted[1:] = cur[:-1]
        neg_r = zo.fr_from_int((R - zo.fr_to_int(r[j])) % R)
        cur = zo.linear_combination([shifted, cur], np.stack([zo.fr_from_int(1), neg_r]))
    d = ctx.to_device(cur)
    ffi.kate_division_device(ctx, [d], [r])
    assert (ctx.to_host(d) == q).all()


@pytest.mark.parametrize("n,count", [(100, 1), (1 << 13, 5), (1 << 16, 20)])
def test_divide_by_linear_vs_oracle(zk, oracle, n, count):
    """independent divisions into fresh outputs (the partial-fraction form the multi-open uses); count > 16 spans two launches"""
    ffi, ctx = zk
    zo = oracle
    srcs = [zo.synth_raw253(7600 + j, n) for j in range(min(count, 3))]
    roots = zo.synth_raw253(7700 + n, count)
    dev = [ctx.to_device(q) for q in srcs]
    outs = ffi.divide_by_linear_device(ctx, [dev[j % len(dev)] for j in range(count)], roots)
    for j in range(count):
        assert (ctx.to_host(outs[j]) == zo.kate_division(srcs[j % len(srcs)], roots[j:j + 1])).all()
    for q, d in zip(srcs, dev):
        assert (ctx.to_host(d) == q).all()          # sources untouched


def test_native_multiopen_golden(zk, oracle):
    """zkhip_shplonk_open (rotation sets, interpolation, partial fractions and the transcript order in C++) against the golden
    commitments, which the generator checked with the verifier equation."""
    import halo2_zkcert_amd.prover as pv

    ffi, ctx = zk
    zo = oracle
    g, F = _case(zo)
    b = pv.GpuBackend(ctx, ffi)
    b.setup(g["k"], 3, H(g["s"]))
    polys = {i: ctx.to_device(F(c)) for i, c in enumerate(g["polys"])}
    queries = [(i, H(pt), H(e)) for i, pt, e in g["queries"]]
    flat = zo.fr_arr_from_ints([e for _, _, e in queries])
    ch = {"shplonk_y": H(g["y"]), "shplonk_v": H(g["v"]), "shplonk_u": H(g["u"])}
    order = []
    pr = b.multiopen(polys, [(i, pt) for i, pt, _ in queries], flat, lambda t: (order.append(t), ch[t])[1], lambda t, pts: order.append(t))
    assert order == ["shplonk_y", "shplonk_v", "shplonk_h1", "shplonk_u", "shplonk_h2"]
    for got, exp in ((pr["h1"], g["h1"]), (pr["h2"], g["h2"])):
        assert zo.affine_to_ints(np.asarray(got[0]).reshape(1, 8))[0] == (H(exp[0]), H(exp[1]))


@pytest.mark.parametrize("rots_a,rots_b,n_sets,max_set,n_points", [
    (list(range(-5, 7)), list(range(-6, 7)), 3, 13, 13),
    # 21 distinct points: more than one launch batch of the per-point divisions (KD_MAX = 16); points of one set only and of several
    (list(range(-9, 10)), [0, 1, 20, 21], 3, 19, 21),
])
def test_native_multiopen_many_rotations(zk, oracle, rots_a, rots_b, n_sets, max_set, n_points):
    """Rotation sets of 12 and 13 points (> 8: the zkevm SHA-256 bit circuit queries its bit columns at many rotations,
    /root/reference/src/sha256_bit_circuit.rs:51-56; upstream's SHPLONK has no limit): zkhip_shplonk_open == the Python SHPLONK on the
    oracle backend (commitments h1, h2), and the verifier's equation under the SRS trapdoor holds.  The library divides once per
    DISTINCT point (partial fractions regrouped by root); the Python SHPLONK and the oracle divide per (set, point) as upstream does."""
    import halo2_zkcert_amd.prover as pv
    import halo2_zkcert_amd.shplonk as sp
    import pyref
    from oracle_backend import OracleBackend

    ffi, ctx = zk
    zo = oracle
    k, n, s = 9, 1 << 9, 0x5EED1234
    omega = pow(pv.ROOT_OF_UNITY, 1 << (28 - k), R)
    x = 0x1234567890ABCDEF1234567890ABCDEF % R
    polys_h = [zo.synth_raw253(7800 + j, n) for j in range(4)]
    pt = lambda r_: x * pow(omega, r_ % (1 << k), R) % R
    queries = []
    for pid, rots in ((0, rots_a), (1, [0, 1]), (2, rots_a), (3, rots_b)):
        for r_ in rots:
            e = zo.fr_to_int(zo.eval_polynomial(polys_h[pid], zo.fr_from_int(pt(r_))))
            queries.append((pid, pt(r_), e))
    ch = {"shplonk_y": 0x1111111122222222 % R, "shplonk_v": 0x3333333344444444555555 % R, "shplonk_u": 0x66666666777777778888888899 % R}
    ob = OracleBackend(4)
    ob.setup(k, 3, s)
    ref = sp.ProverSHPLONK(ob).create_proof({i: q.copy() for i, q in enumerate(polys_h)}, queries, lambda t: ch[t], lambda t, c: None)
    b = pv.GpuBackend(ctx, ffi)
    b.setup(k, 3, s)
    polys = {i: ctx.to_device(q) for i, q in enumerate(polys_h)}
    flat = zo.fr_arr_from_ints([e for _, _, e in queries])
    pr = b.multiopen(polys, [(i, p_) for i, p_, _ in queries], flat, lambda t: ch[t], lambda t, pts: None)
    aff = lambda g_: zo.affine_to_ints(np.asarray(g_[0]).reshape(1, 8))[0]
    assert aff(pr["h1"]) == aff(ref["h1"]) and aff(pr["h2"]) == aff(ref["h2"])
    # the verifier's side, sharing no code with either prover: commitments by the oracle's MSM, equation under the trapdoor
    commits = {i: zo.affine_to_ints(zo.g1_to_affine(zo.best_multiexp(q, ob.g, 4)).reshape(1, 8))[0] for i, q in enumerate(polys_h)}
    sets, supers = [], sorted({p_ for _, p_, _ in queries})
    for pid in range(4):
        pts = sorted(p_ for i, p_, _ in queries if i == pid)
        ev = {p_: e for i, p_, e in queries if i == pid}
        for rs in sets:
            if rs["points"] == pts:
                rs["commitments"].append((pid, [ev[p_] for p_ in pts]))
                break
        else:
            sets.append({"points": pts, "commitments": [(pid, [ev[p_] for p_ in pts])]})
    assert len(sets) == n_sets and max(len(rs["points"]) for rs in sets) == max_set and len(supers) == n_points
    assert pyref.shplonk_verify(commits, sets, supers, ch["shplonk_y"], ch["shplonk_v"], ch["shplonk_u"], aff(pr["h1"]), aff(pr["h2"]), s)
