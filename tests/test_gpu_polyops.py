"""GPU parity: grand products, batch inversion and evaluation at a point vs golden vectors and the oracle."""
import numpy as np
import pytest

from util import H, load

pytestmark = pytest.mark.gpu
R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001


def test_golden(zk, oracle):
    ffi, ctx = zk
    zo = oracle
    g = load("products.json")
    k, bf = g["k"], g["bf"]
    F = lambda xs: zo.fr_arr_from_ints([H(x) for x in xs])
    values = [ctx.to_device(F(c)) for c in g["values"]]
    sigmas = [ctx.to_device(F(c)) for c in g["sigmas"]]
    beta, gamma = zo.fr_from_int(H(g["beta"])), zo.fr_from_int(H(g["gamma"]))
    blind = ctx.to_device(np.concatenate([F(b) for b in g["blinding"]]))
    zs = ffi.permutation_products_device(ctx, k, values, sigmas, g["chunk_len"], beta, gamma, bf, blind)
    for z, exp in zip(zs, g["z"]):
        assert zo.fr_arr_to_ints(ctx.to_host(z)) == [H(x) for x in exp]
    L = g["lookup"]
    lz = ffi.lookup_product_device(ctx, k, *[ctx.to_device(F(L[n])) for n in ("cin", "ctab", "pin", "ptab")], beta, gamma, bf,
                                   ctx.to_device(F(L["blinding"])))
    assert zo.fr_arr_to_ints(ctx.to_host(lz)) == [H(x) for x in L["z"]]
    bi = g["batch_invert"]
    col = ctx.to_device(F(bi["input"]))
    ffi.batch_invert_device(ctx, col)
    assert zo.fr_arr_to_ints(ctx.to_host(col)) == [H(x) for x in bi["output"]]
    ev = g["evals"]
    out = ffi.eval_polynomials_device(ctx, values, zo.fr_from_int(H(ev["x"])))
    assert zo.fr_arr_to_ints(ctx.to_host(out)) == [H(x) for x in ev["values"]]


@pytest.mark.parametrize("k,ncols,chunk,bf", [(9, 6, 2, 6), (12, 5, 3, 5), (13, 1, 2, 3)])
def test_products_vs_oracle(zk, oracle, k, ncols, chunk, bf):
    ffi, ctx = zk
    zo = oracle
    n = 1 << k
    vals = [zo.synth_raw253(5000 + k * 10 + j, n) for j in range(ncols)]
    sigs = [zo.synth_raw253(5100 + k * 10 + j, n) for j in range(ncols)]
    vals[0][5] = 0                       # a zero term: the fraction's numerator can vanish
    beta, gamma = zo.synth_raw253(5200, 2)
    nsets = -(-ncols // chunk)
    blind = zo.synth_raw253(5300 + k, nsets * bf)
    exp = zo.permutation_products(k, vals, sigs, chunk, beta, gamma, bf, blind.reshape(nsets, bf, 4))
    got = ffi.permutation_products_device(ctx, k, [ctx.to_device(v) for v in vals], [ctx.to_device(v) for v in sigs], chunk, beta, gamma,
                                          bf, ctx.to_device(blind))
    for a, b in zip(got, exp):
        assert (ctx.to_host(a) == b).all()
    cols = [zo.synth_raw253(5400 + k * 10 + j, n) for j in range(4)]
    lb = zo.synth_raw253(5500 + k, bf)
    expl = zo.lookup_product(k, *cols, beta, gamma, bf, lb)
    gotl = ffi.lookup_product_device(ctx, k, *[ctx.to_device(c) for c in cols], beta, gamma, bf, ctx.to_device(lb))
    assert (ctx.to_host(gotl) == expl).all()


@pytest.mark.parametrize("n", [1, 7, 2048, 2049, 100000])
def test_batch_invert_and_eval(zk, oracle, n):
    ffi, ctx = zk
    zo = oracle
    a = zo.synth_raw253(6000 + n, n)
    if n > 3:
        a[3] = 0
    col = ctx.to_device(a)
    ffi.batch_invert_device(ctx, col)
    assert (ctx.to_host(col) == zo.batch_invert(a)).all()
    x = zo.synth_raw253(6100, 1)[0]
    polys = [zo.synth_raw253(6200 + n + j, n) for j in range(3)]
    out = ffi.eval_polynomials_device(ctx, [ctx.to_device(q) for q in polys], x)
    assert (ctx.to_host(out) == zo.eval_polynomials(polys, x)).all()


def test_full_size_permutation_telescopes(zk, oracle):
    """2^17 rows (BASELINE configs[1] size): with sigma = the identity permutation (sigma_j(w^i) = delta^j w^i) every
    fraction is 1, and with a genuine non-trivial permutation of equal-valued cells the product telescopes: in both
    cases z = 1 on every non-blinding row — a size-independent check of the whole numerator/denominator/scan chain."""
    ffi, ctx = zk
    zo = oracle
    k, bf, ncols, chunk = 17, 5, 4, 2      # n - bf - 1 even: the chained row closes a swapped pair
    n = 1 << k
    w = zo.root_of_unity(k)
    zeta, delta = zo.fr_constants()
    # identity sigmas on the host: delta^j * w^i
    wi = np.zeros((n, 4), dtype=np.uint64)
    cur = zo.fr_from_int(1)
    step = w
    # build w^i with the GPU: coeff_to_lagrange of the polynomial X evaluates X at every w^i
    xpoly = np.zeros((n, 4), dtype=np.uint64)
    xpoly[1] = zo.fr_from_int(1)
    dom = ffi.EvaluationDomain(ctx, 3, k)
    t = ctx.to_device(xpoly)
    dom.coeff_to_lagrange_device([t])
    wi = ctx.to_host(t)
    assert (wi[1] == w).all() and (wi[0] == zo.fr_from_int(1)).all()
    dj = zo.fr_from_int(1)
    sigmas = []
    import ctypes as C

    def scale(col, s):
        out = col.copy()
        L = zo.lib()
        for i in range(0, n, 1):
            L.zko_fr_mul(C.c_void_p(col.ctypes.data + 32 * i), zo.p(s), C.c_void_p(out.ctypes.data + 32 * i))
        return out

    for j in range(ncols):
        sigmas.append(scale(wi, dj))
        dj = zo._binary("zko_fr_mul", dj, delta)
    # permutation: swap rows 2i <-> 2i+1 of column 0 and give both cells the same value
    vals = [zo.synth_raw253(7000 + j, n) for j in range(ncols)]
    vals[0][1::2] = vals[0][0::2]
    s0 = sigmas[0].copy()
    s0[0::2], s0[1::2] = sigmas[0][1::2], sigmas[0][0::2]
    sigmas[0] = s0
    beta, gamma = zo.synth_raw253(7100, 2)
    nsets = ncols // chunk
    blind = zo.synth_raw253(7200, nsets * bf)
    zs = ffi.permutation_products_device(ctx, k, [ctx.to_device(v) for v in vals], [ctx.to_device(s) for s in sigmas], chunk, beta, gamma,
                                         bf, ctx.to_device(blind))
    one = zo.fr_from_int(1)
    z0 = ctx.to_host(zs[0])
    # swapped pairs: z returns to 1 after every pair
    assert (z0[0:n - bf:2] == one).all()
    z1 = ctx.to_host(zs[1])
    assert (z1[: n - bf] == one).all()          # identity sigmas: constant 1 (chained from z0's last kept row = 1)
    assert (z1[n - bf:] == blind.reshape(nsets, bf, 4)[1]).all()
    dom.free()


def _lookup_columns(zo, k, bf, seed, distinct, dup_table=False):
    """A satisfiable (input, table) pair: the table holds `distinct` different values (the rest repeats of them or of a
    filler), the input draws from the table with a skewed distribution."""
    n = 1 << k
    u = n - (bf + 1)
    rng = np.random.default_rng(seed)
    pool = zo.synth_raw253(seed, distinct)
    for i in range(distinct):
        pool[i][3] &= (1 << 60) - 1      # canonical (< r)
    tab = np.empty((n, 4), dtype=np.uint64)
    tab[:distinct] = pool
    tab[distinct:] = pool[rng.integers(0, distinct if dup_table else 1, n - distinct)]
    tab[:u] = tab[:u][rng.permutation(u)] if distinct <= u else tab[:u]
    present = np.unique(tab[:u], axis=0)
    idx = np.minimum((rng.random(n) ** 3 * len(present)).astype(np.int64), len(present) - 1)
    inp = present[idx]
    return inp, tab


@pytest.mark.parametrize("k,bf,distinct,dup", [(4, 5, 3, True), (8, 5, 100, False), (11, 5, 1500, True), (12, 6, 4000, False), (14, 5, 9000, True)])
def test_permute_expression_pair_vs_oracle(zk, oracle, k, bf, distinct, dup):
    ffi, ctx = zk
    zo = oracle
    inp, tab = _lookup_columns(zo, k, bf, 6100 + k, min(distinct, (1 << k) - bf - 1), dup)
    bi, bt = zo.synth_raw253(6200 + k, bf + 1), zo.synth_raw253(6300 + k, bf + 1)
    ea, es = zo.permute_expression_pair(k, bf, inp, tab, bi, bt)
    ga, gs = ffi.permute_expression_pair_device(ctx, k, bf, ctx.to_device(inp), ctx.to_device(tab), ctx.to_device(bi), ctx.to_device(bt))
    assert (ctx.to_host(ga) == ea).all()
    assert (ctx.to_host(gs) == es).all()


def test_permute_expression_pair_golden_and_failure(zk, oracle):
    ffi, ctx = zk
    zo = oracle
    g = load("products.json")
    pm = g["permute"]
    F = lambda xs: zo.fr_arr_from_ints([H(x) for x in xs])
    dev = lambda name: ctx.to_device(F(pm[name]))
    ga, gs = ffi.permute_expression_pair_device(ctx, g["k"], g["bf"], dev("input"), dev("table"), dev("blind_in"), dev("blind_tab"))
    assert zo.fr_arr_to_ints(ctx.to_host(ga)) == [H(x) for x in pm["permuted_input"]]
    assert zo.fr_arr_to_ints(ctx.to_host(gs)) == [H(x) for x in pm["permuted_table"]]
    bad = F(pm["input"])
    bad[1] = zo.fr_from_int(0x1234567)   # not a table value
    with pytest.raises(ffi.ConstraintSystemFailure):
        ffi.permute_expression_pair_device(ctx, g["k"], g["bf"], ctx.to_device(bad), dev("table"), dev("blind_in"), dev("blind_tab"))


def test_permute_expression_pair_properties_large(zk, oracle):
    """k = 17 (the RSA circuit's size): multiset equality with the inputs and the argument's row rule, checked on the host
    with numpy on the canonical values."""
    ffi, ctx = zk
    zo = oracle
    k, bf = 17, 5
    n, u = 1 << k, (1 << k) - 6
    inp, tab = _lookup_columns(zo, k, bf, 6400, 1 << 16, True)
    bi, bt = zo.synth_raw253(6500, bf + 1), zo.synth_raw253(6600, bf + 1)
    ga, gs = ffi.permute_expression_pair_device(ctx, k, bf, ctx.to_device(inp), ctx.to_device(tab), ctx.to_device(bi), ctx.to_device(bt))
    ga, gs = ctx.to_host(ga), ctx.to_host(gs)
    assert (ga[u:] == bi).all() and (gs[u:] == bt).all()
    key = lambda a: [tuple(r) for r in a]
    assert sorted(key(ga[:u])) == sorted(key(inp[:u]))
    assert sorted(key(gs[:u])) == sorted(key(tab[:u]))
    same_as_table = (ga[:u] == gs[:u]).all(axis=1)
    same_as_prev = np.concatenate([[False], (ga[1:u] == ga[:u - 1]).all(axis=1)])
    assert (same_as_table | same_as_prev).all() and same_as_table[0]
    ea, es = zo.permute_expression_pair(k, bf, inp, tab, bi, bt)
    assert (ga == ea).all() and (gs == es).all()
