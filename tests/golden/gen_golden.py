#!/usr/bin/env python3
"""Generates the golden fixtures in this directory from oracle/pyref.py (first-principles big ints).

    python tests/golden/gen_golden.py

The reference has no vectors for this path (parity unpinned), so these are
mathematically determined values (canonical integers), plus two externally known anchors:
the EIP-196 value of 2*G1 and the survey's verified roots of unity.
All field values are canonical (non-Montgomery) hex strings.
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import pyref as P  # noqa: E402

hx = lambda x: format(x, "x")


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, indent=0, separators=(",", ":"))
    print(name, os.path.getsize(os.path.join(HERE, name)), "bytes")


def field_vectors():
    out = {}
    for name, m, seed in (("fr", P.R, 11), ("fq", P.P, 12)):
        xs = [0, 1, 2, m - 1, m - 2, (1 << 253) - 1, (1 << 128) + 5] + [P.synth_raw253(seed, i) for i in range(9)]
        cases = []
        for i in range(len(xs)):
            a, b = xs[i], xs[(i * 7 + 3) % len(xs)]
            cases.append(dict(a=hx(a), b=hx(b), mul=hx(a * b % m), add=hx((a + b) % m), sub=hx((a - b) % m),
                              inv=hx(pow(a, -1, m) if a else 0), mont=hx(P.to_mont(a, m))))
        out[name] = cases
    out["constants"] = dict(
        root_of_unity=hx(P.ROOT_OF_UNITY), delta=hx(P.DELTA), zeta=hx(P.ZETA),
        omega={str(k): hx(P.omega_for(k)) for k in (1, 2, 4, 10, 15, 17, 19, 22, 24, 28)},
        # SURVEY.md §8 'Verified constants' — computed there with sympy, repeated as an external anchor
        survey_omega_17="304cd1e79cfa5b0f054e981a27ed7706e7ea6b06a7f266ef8db819c179c2c3ea",
        survey_omega_19="cf1526aaafac6bacbb67d11a4077806b123f767e4b0883d14cc0193568fc082")
    return out


def g1_vectors():
    ks = [1, 2, 3, 5, 7, 255, 256, (1 << 64) + 1, (1 << 128) - 1, P.R - 1, P.R - 2] + [P.synth_raw253(21, i) for i in range(5)]
    muls = []
    for k in ks:
        a = P.to_affine(P.scalar_mul(k, P.G1_GEN))
        assert P.on_curve(a)
        muls.append(dict(k=hx(k), x=hx(a[0]), y=hx(a[1]), compressed=P.compress(a).hex()))
    adds = []
    pts = [P.scalar_mul(k, P.G1_GEN) for k in (3, 5, 5, P.R - 5, 0, 9)]
    for i in range(len(pts)):
        for j in range(len(pts)):
            s = P.to_affine(P.jac_add(pts[i], pts[j]))
            adds.append(dict(i=i, j=j, x=hx(s[0]), y=hx(s[1])))
    return dict(
        mul_gen=muls,
        add_operands=[dict(zip("xy", map(hx, P.to_affine(p)))) for p in pts],
        adds=adds,
        # EIP-196 / alt_bn128 published value of 2*(1,2)
        eip196_2g=dict(x="30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd3",
                       y="15ed738c0e0a7c92e7845f96b2ae9c0a68a6a449e3538fc7ff3ebf7a5a18a2c4"),
        identity_compressed=P.compress((0, 0)).hex())


def msm_vectors():
    cases = []

    def case(name, scalars, points):
        res = P.msm_naive(scalars, points)
        cases.append(dict(name=name, scalars=[hx(s) for s in scalars], points=[[hx(x), hx(y)] for x, y in points],
                          result=[hx(res[0]), hx(res[1])]))

    def pt(i, seed=31):
        return P.to_affine(P.scalar_mul(P.synth_raw253(seed, i), P.G1_GEN))

    g = (1, 2)
    case("n1_one", [1], [g])
    case("n1_zero", [0], [g])
    case("n1_rminus1", [P.R - 1], [g])
    case("n2_cancel", [5, 5], [pt(0), (pt(0)[0], (-pt(0)[1]) % P.P)])
    case("n3_dup_points", [7, 7, 9], [pt(1), pt(1), pt(2)])            # P + P inside one bucket
    case("n3_identity_base", [3, 4, 5], [pt(3), (0, 0), pt(4)])
    edge = [0, 1, 2, P.R - 1, P.R - 2, (1 << 13) - 1, 1 << 13, (1 << 16) - 1, 1 << 16, (1 << 15), (1 << 253) - 1,
            (1 << 128), (1 << 64) - 1, 0xFFFF0000FFFF0000FFFF, 3, 1, 1]
    case("n17_edge_scalars", edge, [pt(10 + i) for i in range(17)])
    n = 256
    sc = [P.from_mont(P.synth_raw253(32, i), P.R) for i in range(n)]
    for i in range(0, n, 16):
        sc[i] = i % 3          # small / zero scalars as in real witnesses
    case("n256_mixed", sc, [pt(100 + i) for i in range(n)])
    # seed-defined case: scalars[i] = from_mont(raw253(seed_s, i)); points[i] = [raw253(seed_p, i)] G
    n = 1024
    sc = [P.from_mont(P.synth_raw253(41, i), P.R) for i in range(n)]
    pts = [P.to_affine(P.scalar_mul(P.synth_raw253(42, i), P.G1_GEN)) for i in range(n)]
    res = P.msm_naive(sc, pts)
    seeded = dict(n=n, seed_scalars=41, seed_points=42, result=[hx(res[0]), hx(res[1])],
                  point0=[hx(pts[0][0]), hx(pts[0][1])], point_last=[hx(pts[-1][0]), hx(pts[-1][1])])
    return dict(cases=cases, seeded=seeded)


def digest(vals):
    h = hashlib.sha256()
    for v in vals:
        h.update(v.to_bytes(32, "little"))
    return h.hexdigest()


def ntt_vectors():
    out = []
    for k in (1, 2, 3, 6):
        n = 1 << k
        a = [P.from_mont(P.synth_raw253(50 + k, i), P.R) for i in range(n)]
        w = P.omega_for(k)
        f = P.dft_naive(a, w)
        assert P.fft(a, w) == f
        out.append(dict(k=k, input=[hx(x) for x in a], output=[hx(x) for x in f]))
    big = []
    for k in (10, 11, 13):
        n = 1 << k
        a = [P.from_mont(P.synth_raw253(50 + k, i), P.R) for i in range(n)]
        w = P.omega_for(k)
        f = P.fft(a, w)
        inv = P.ifft(a, w)
        big.append(dict(k=k, seed=50 + k, fft_sha256=digest(f), ifft_sha256=digest(inv),
                        fft_first=[hx(x) for x in f[:4]], ifft_first=[hx(x) for x in inv[:4]]))
    return dict(small=out, seeded=big)


def domain_vectors():
    out = []
    for (j, k) in ((4, 4), (3, 3), (6, 3), (9, 2)):
        d = P.Domain(j, k)
        coeffs = [P.from_mont(P.synth_raw253(60 + k + j, i), P.R) for i in range(d.n)]
        ext = d.coeff_to_extended(coeffs)
        for i in (0, 1, d.extended_n - 1):
            assert ext[i] == P.poly_eval(coeffs, d.coset_point(i))
        extin = [P.from_mont(P.synth_raw253(70 + k + j, i), P.R) for i in range(d.extended_n)]
        l0, ll, la = P.l_cosets(d, 3 if k > 2 else 1)
        out.append(dict(j=j, k=k, extended_k=d.extended_k, blinding_factors=3 if k > 2 else 1,
                        coeffs=[hx(x) for x in coeffs],
                        lagrange_to_coeff=[hx(x) for x in d.lagrange_to_coeff(coeffs)],
                        coeff_to_extended=[hx(x) for x in ext],
                        extended_in=[hx(x) for x in extin],
                        extended_to_coeff=[hx(x) for x in d.extended_to_coeff(extin)],
                        divide_by_vanishing=[hx(x) for x in d.divide_by_vanishing_poly(extin)],
                        t_evaluations=[hx(x) for x in d.t_evaluations],
                        l0=[hx(x) for x in l0], l_last=[hx(x) for x in ll], l_active=[hx(x) for x in la]))
    return out


def evalh_vectors():
    """A small non-satisfying circuit exercising every term kind of evaluate_h."""
    k, degree, bf = 4, 4, 5
    d = P.Domain(degree, k)
    n = d.n
    A0 = lambda r: ["advice", 0, r]
    A1 = lambda r: ["advice", 1, r]
    F0 = lambda r: ["fixed", 0, r]
    F1 = lambda r: ["fixed", 1, r]
    I0 = lambda r: ["instance", 0, r]
    gates = [
        # halo2-lib vertical gate: q * (a + b*c - d)
        ["prod", F0(0), ["sum", ["sum", A0(0), ["prod", A0(1), A0(2)]], ["neg", A0(3)]]],
        # constants, scaling, challenge, negative rotation, instance, x*x and 2*x special cases
        ["sum", ["scaled", ["prod", A1(0), A1(0)], 7], ["neg", ["prod", ["const", 2], A1(-1)]]],
        ["prod", ["sum", I0(0), ["challenge", 0]], ["sum", F1(1), ["const", P.R - 3]]],
        ["sum", ["prod", ["const", 1], A0(0)], ["prod", ["const", 0], A1(2)]],
        ["neg", ["const", 5]],
    ]
    lookups = [([["prod", F0(0), A1(0)], A0(1)], [F1(0), ["sum", F1(0), ["const", 1]]])]
    perm_columns = [["advice", 0], ["fixed", 1], ["instance", 0]]
    seed = [80]

    def rnd_poly(m=n):
        seed[0] += 1
        return [P.from_mont(P.synth_raw253(seed[0], i), P.R) for i in range(m)]

    polys = dict(fixed=[rnd_poly(), rnd_poly()], advice=[rnd_poly(), rnd_poly()], instance=[rnd_poly()],
                 sigma=[rnd_poly() for _ in perm_columns], perm_z=[rnd_poly(), rnd_poly()],
                 lookup_z=[rnd_poly()], lookup_a=[rnd_poly()], lookup_s=[rnd_poly()])
    cosets = {kk: [d.coeff_to_extended(p) for p in v] for kk, v in polys.items()}
    cosets["l0"], cosets["l_last"], cosets["l_active"] = P.l_cosets(d, bf)
    ch = dict(beta=P.synth_raw253(90, 0) % P.R, gamma=P.synth_raw253(90, 1) % P.R, theta=P.synth_raw253(90, 2) % P.R,
              y=P.synth_raw253(90, 3) % P.R, challenges=[P.synth_raw253(90, 4) % P.R])

    def tup(e):
        return tuple(tup(x) if isinstance(x, list) else x for x in e)

    cs = dict(gates=[tup(g) for g in gates], lookups=[([tup(e) for e in i], [tup(e) for e in t]) for i, t in lookups],
              perm_columns=[tuple(c) for c in perm_columns], degree=degree, blinding_factors=bf)
    h = P.evaluate_h_direct(d, cs, cosets, ch)
    return dict(k=k, degree=degree, blinding_factors=bf, gates=gates, lookups=[list(l) for l in lookups],
                perm_columns=perm_columns, polys={kk: [[hx(x) for x in p] for p in v] for kk, v in polys.items()},
                challenges={kk: ([hx(x) for x in v] if isinstance(v, list) else hx(v)) for kk, v in ch.items()},
                h=[hx(x) for x in h])


def products_vectors():
    k, bf, chunk = 4, 3, 2
    n = 1 << k
    seed = [700]

    def col():
        seed[0] += 1
        return [P.from_mont(P.synth_raw253(seed[0], i), P.R) for i in range(n)]

    values = [col() for _ in range(5)]
    sigmas = [col() for _ in range(5)]
    beta, gamma = P.synth_raw253(710, 0) % P.R, P.synth_raw253(710, 1) % P.R
    nsets = 3
    blind = [[P.synth_raw253(720 + s, t) % P.R for t in range(bf)] for s in range(nsets)]
    z = P.permutation_products(k, values, sigmas, chunk, beta, gamma, bf, blind)
    cin, ctab, pin, ptab = col(), col(), col(), col()
    lblind = [P.synth_raw253(730, t) % P.R for t in range(bf)]
    lz = P.lookup_product(k, cin, ctab, pin, ptab, beta, gamma, bf, lblind)
    # lookup permutation: a small table with repeats, inputs drawn from it (plus values > 2^128 to exercise the full compare)
    import random
    rnd = random.Random(5)
    u = n - (bf + 1)
    base_vals = [0, 1, 2, 3, 5, (1 << 130) + 7, (1 << 130) + 8, P.R - 1]
    tab = [rnd.choice(base_vals) for _ in range(u)]
    inp = [rnd.choice(tab) for _ in range(u)]
    tab_full = tab + [P.synth_raw253(750, i) % P.R for i in range(bf + 1)]
    inp_full = inp + [P.synth_raw253(751, i) % P.R for i in range(bf + 1)]
    bi = [P.synth_raw253(752, i) % P.R for i in range(bf + 1)]
    bt = [P.synth_raw253(753, i) % P.R for i in range(bf + 1)]
    pa, ps = P.permute_expression_pair(k, bf, inp_full, tab_full, bi, bt)
    permute = dict(input=[hx(v) for v in inp_full], table=[hx(v) for v in tab_full], blind_in=[hx(v) for v in bi], blind_tab=[hx(v) for v in bt],
                   permuted_input=[hx(v) for v in pa], permuted_table=[hx(v) for v in ps])
    inv_in = [0, 1, 2, P.R - 1] + col()[:6]
    x = P.synth_raw253(740, 0) % P.R
    return dict(k=k, bf=bf, chunk_len=chunk, beta=hx(beta), gamma=hx(gamma),
                values=[[hx(v) for v in c] for c in values], sigmas=[[hx(v) for v in c] for c in sigmas],
                blinding=[[hx(v) for v in b] for b in blind], z=[[hx(v) for v in c] for c in z],
                lookup=dict(cin=[hx(v) for v in cin], ctab=[hx(v) for v in ctab], pin=[hx(v) for v in pin], ptab=[hx(v) for v in ptab],
                            blinding=[hx(v) for v in lblind], z=[hx(v) for v in lz]),
                batch_invert=dict(input=[hx(v) for v in inv_in], output=[hx(pow(v, -1, P.R) if v else 0) for v in inv_in]),
                evals=dict(x=hx(x), values=[hx(P.poly_eval(c, x)) for c in values]), permute=permute)


def shplonk_vectors():
    """SHPLONK prover polynomials h(X), h'(X) for six polynomials opened at four different point sets (k = 5), the
    commitments under the SRS trapdoor, and the verifier's verdict (pyref.shplonk_verify: an independent equation)."""
    k, s = 5, 0x1D5C0FFEE
    n = 1 << k
    polys = {i: [P.synth_raw253(800 + i, j) % P.R for j in range(n)] for i in range(6)}
    x = P.synth_raw253(810, 0) % P.R
    w = P.omega_for(k)
    pts = {0: [x], 1: [x, x * w % P.R], 2: [x * w % P.R, x], 3: [x, x * w % P.R, x * pow(w, n - 3, P.R) % P.R], 4: [x],
           5: [x, x * pow(w, n - 1, P.R) % P.R]}
    queries = [(i, pt, P.poly_eval(polys[i], pt)) for i in [0, 1, 3, 2, 5, 4] for pt in pts[i]]
    y, v, u = (P.synth_raw253(811, j) % P.R for j in range(3))
    rs, sp = P.construct_intermediate_sets(queries)
    h = P.shplonk_quotient(polys, rs, y, v, n)
    f = P.shplonk_linearisation(polys, rs, sp, y, v, u, h, n)
    G = P.from_affine((1, 2))
    com = lambda q: P.to_affine(P.scalar_mul(P.poly_eval(q, s), G))
    C = {i: com(polys[i]) for i in polys}
    h1, h2 = com(h), com(f)
    assert P.shplonk_verify(C, rs, sp, y, v, u, h1, h2, s)
    # primitives: one linear combination with a low-degree correction, one division by three roots
    cf = [P.synth_raw253(812, j) % P.R for j in range(6)]
    low = [P.synth_raw253(813, j) % P.R for j in range(3)]
    lc = [(sum(cf[i] * polys[i][d] for i in range(6)) - (low[d] if d < 3 else 0)) % P.R for d in range(n)]
    roots = [3, x, P.R - 5]
    prod = list(polys[0][:n - 3])
    for r in roots:
        prod = [(a - r * b_) % P.R for a, b_ in zip([0] + prod, prod + [0])]
    return dict(k=k, s=hx(s), polys=[[hx(c) for c in polys[i]] for i in range(6)],
                queries=[[i, hx(pt), hx(e)] for i, pt, e in queries], y=hx(y), v=hx(v), u=hx(u),
                set_sizes=[len(r["points"]) for r in rs], set_members=[[c[0] for c in r["commitments"]] for r in rs],
                h=[hx(c) for c in h], h_prime=[hx(c) for c in f],
                commitments=[[hx(C[i][0]), hx(C[i][1])] for i in range(6)], h1=[hx(h1[0]), hx(h1[1])], h2=[hx(h2[0]), hx(h2[1])],
                lincomb=dict(coeffs=[hx(c) for c in cf], low=[hx(c) for c in low], out=[hx(c) for c in lc]),
                division=dict(dividend=[hx(c) for c in prod], roots=[hx(r) for r in roots], quotient=[hx(c) for c in polys[0][:n - 3]] + ["0"] * 3))


def poseidon_vectors():
    """Poseidon (t = 3, R_F = 8, R_P = 57 over BN254 Fr) from pyref's Grain-generated parameters: the published permutation vector, the
    first / last round constants and the MDS matrix, permutations of edge states, and sponge sequences (empty / short / exact /
    ragged buffers, repeated squeezes)."""
    import random

    rc, mds = P.poseidon_spec()
    rnd = random.Random(0xC0FFEE)
    perms = []
    for st in ([0, 0, 0], [1 << 64, 0, 0], [P.R - 1, P.R - 2, P.R - 3], [rnd.randrange(P.R) for _ in range(3)]):
        perms.append({"in": [hx(v) for v in st], "out": [hx(v) for v in P.poseidon_permute(st)]})
    sponge = []
    for lens in ([0], [1], [2], [3, 0, 2], [5, 4, 1]):
        sp, absorb, sq = P.PoseidonSponge(), [], []
        for ln in lens:
            b = [rnd.randrange(P.R) for _ in range(ln)]
            sp.update(b)
            absorb.append([hx(v) for v in b])
            sq.append(hx(sp.squeeze()))
        sponge.append(dict(absorb=absorb, squeezed=sq))
    assert P.poseidon_permute([0, 1, 2]) == P.POSEIDON_KAT
    return dict(source="oracle/pyref.py (Grain LFSR parameters, t=3, R_F=8, R_P=57, BN254 Fr); kat = hadeshash test_vectors.txt poseidonperm_x5_254_3",
                kat_poseidonperm_x5_254_3=[hx(v) for v in P.POSEIDON_KAT],
                round_constants_first_last=[hx(v) for v in rc[0] + rc[-1]], mds=[hx(v) for row in mds for v in row],
                permutations=perms, sponge=sponge)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "poseidon":
        dump("poseidon.json", poseidon_vectors())
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "products":
        dump("products.json", products_vectors())
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "shplonk":
        dump("shplonk.json", shplonk_vectors())
        sys.exit(0)
    dump("field.json", field_vectors())
    dump("g1.json", g1_vectors())
    dump("msm.json", msm_vectors())
    dump("ntt.json", ntt_vectors())
    dump("domain.json", domain_vectors())
    dump("evalh.json", evalh_vectors())
    dump("products.json", products_vectors())
    dump("shplonk.json", shplonk_vectors())
