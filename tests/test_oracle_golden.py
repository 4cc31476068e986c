"""CPU: the C restatement (oracle/zkoracle.c) against the first-principles golden vectors."""
import ctypes as C
import hashlib

import numpy as np
import pytest

from util import H, evalh_case, load


def test_field_ops(oracle):
    zo = oracle
    g = load("field.json")
    for name in ("fr", "fq"):
        frm = getattr(zo, f"{name}_from_int")
        to = getattr(zo, f"{name}_to_int")
        for c in g[name]:
            a, b = frm(H(c["a"])), frm(H(c["b"]))
            assert zo.limbs_to_int(a) == H(c["mont"])
            assert to(zo._binary(f"zko_{name}_mul", a, b)) == H(c["mul"])
            assert to(zo._binary(f"zko_{name}_add", a, b)) == H(c["add"])
            assert to(zo._binary(f"zko_{name}_sub", a, b)) == H(c["sub"])
            assert to(zo._unary(f"zko_{name}_inv", a)) == H(c["inv"])


def test_constants(oracle):
    zo = oracle
    c = load("field.json")["constants"]
    for k, v in c["omega"].items():
        assert zo.fr_to_int(zo.root_of_unity(int(k))) == H(v)
    assert H(c["omega"]["17"]) == H(c["survey_omega_17"]) and H(c["omega"]["19"]) == H(c["survey_omega_19"])
    z, d = zo.fr_constants()
    assert zo.fr_to_int(z) == H(c["zeta"]) and zo.fr_to_int(d) == H(c["delta"])


def test_g1(oracle):
    zo = oracle
    g = load("g1.json")
    for m in g["mul_gen"]:
        aff = zo.g1_mul_gen(zo.fr_from_int(H(m["k"])))
        assert zo.affine_to_ints(aff)[0] == (H(m["x"]), H(m["y"]))
        assert zo.g1_to_bytes(aff).hex() == m["compressed"]
        assert zo.lib().zko_g1_is_on_curve(zo.p(aff)) == 1
    two = [m for m in g["mul_gen"] if m["k"] == "2"][0]
    assert (two["x"], two["y"]) == (g["eip196_2g"]["x"], g["eip196_2g"]["y"])
    ops = zo.affine_from_ints([(H(o["x"]), H(o["y"])) for o in g["add_operands"]])
    jac = []
    for a in ops:
        j = zo.new(12)
        zo.lib().zko_g1_from_affine(zo.p(a), zo.p(j))
        jac.append(j)
    for c in g["adds"]:
        o = zo.new(12)
        zo.lib().zko_g1_add(zo.p(jac[c["i"]]), zo.p(jac[c["j"]]), zo.p(o))
        assert zo.affine_to_ints(zo.g1_to_affine(o))[0] == (H(c["x"]), H(c["y"]))
        o2 = zo.new(12)
        zo.lib().zko_g1_add_mixed(zo.p(jac[c["i"]]), zo.p(ops[c["j"]]), zo.p(o2))
        assert zo.affine_to_ints(zo.g1_to_affine(o2))[0] == (H(c["x"]), H(c["y"]))
    assert zo.g1_to_bytes(zo.new(8)).hex() == g["identity_compressed"]


@pytest.mark.parametrize("threads", [1, 3, 8])
def test_msm_cases(oracle, threads):
    zo = oracle
    for c in load("msm.json")["cases"]:
        sc = zo.fr_arr_from_ints([H(s) for s in c["scalars"]])
        pts = zo.affine_from_ints([(H(x), H(y)) for x, y in c["points"]])
        res = zo.g1_to_affine(zo.best_multiexp(sc, pts, threads))
        assert zo.affine_to_ints(res)[0] == (H(c["result"][0]), H(c["result"][1])), c["name"]


def test_msm_seeded_and_synth(oracle):
    zo = oracle
    s = load("msm.json")["seeded"]
    sc = zo.synth_raw253(s["seed_scalars"], s["n"])              # raw limbs ARE the Montgomery form
    pts = zo.fixed_base_mul(zo.fr_arr_from_ints(zo.arr_to_ints(zo.synth_raw253(s["seed_points"], s["n"]))), threads=8)
    assert zo.affine_to_ints(pts[0])[0] == (H(s["point0"][0]), H(s["point0"][1]))
    assert zo.affine_to_ints(pts[-1])[0] == (H(s["point_last"][0]), H(s["point_last"][1]))
    res = zo.g1_to_affine(zo.best_multiexp(sc, pts, 8))
    assert zo.affine_to_ints(res)[0] == (H(s["result"][0]), H(s["result"][1]))


def _digest(ints):
    h = hashlib.sha256()
    for v in ints:
        h.update(v.to_bytes(32, "little"))
    return h.hexdigest()


@pytest.mark.parametrize("threads", [1, 8])
def test_fft(oracle, threads):
    zo = oracle
    g = load("ntt.json")
    for c in g["small"]:
        a = zo.fr_arr_from_ints([H(x) for x in c["input"]])
        out = zo.best_fft(a, zo.root_of_unity(c["k"]), c["k"], threads)
        assert zo.fr_arr_to_ints(out) == [H(x) for x in c["output"]]
    for c in g["seeded"]:
        n = 1 << c["k"]
        a = zo.synth_raw253(c["seed"], n)
        out = zo.fr_arr_to_ints(zo.best_fft(a, zo.root_of_unity(c["k"]), c["k"], threads))
        assert out[:4] == [H(x) for x in c["fft_first"]]
        assert _digest(out) == c["fft_sha256"]
        dom = zo.Domain(3, c["k"])
        inv = zo.fr_arr_to_ints(dom.lagrange_to_coeff(a, threads))
        assert inv[:4] == [H(x) for x in c["ifft_first"]]
        assert _digest(inv) == c["ifft_sha256"]


def test_domain(oracle):
    zo = oracle
    for c in load("domain.json"):
        dom = zo.Domain(c["j"], c["k"])
        assert dom.extended_k == c["extended_k"]
        coeffs = zo.fr_arr_from_ints([H(x) for x in c["coeffs"]])
        assert zo.fr_arr_to_ints(dom.lagrange_to_coeff(coeffs)) == [H(x) for x in c["lagrange_to_coeff"]]
        assert zo.fr_arr_to_ints(dom.coeff_to_extended(coeffs, 2)) == [H(x) for x in c["coeff_to_extended"]]
        ext = zo.fr_arr_from_ints([H(x) for x in c["extended_in"]])
        assert zo.fr_arr_to_ints(dom.extended_to_coeff(ext, 2)) == [H(x) for x in c["extended_to_coeff"]]
        assert zo.fr_arr_to_ints(dom.divide_by_vanishing_poly(ext)) == [H(x) for x in c["divide_by_vanishing"]]
        l0, ll, la = dom.l_cosets(c["blinding_factors"])
        assert zo.fr_arr_to_ints(l0) == [H(x) for x in c["l0"]]
        assert zo.fr_arr_to_ints(ll) == [H(x) for x in c["l_last"]]
        assert zo.fr_arr_to_ints(la) == [H(x) for x in c["l_active"]]


@pytest.mark.parametrize("threads", [1, 4])
def test_evaluate_h(oracle, threads):
    zo = oracle
    dom, kw, expect = evalh_case(zo)
    pack = zo.EvalhPack()
    pack.build(**kw)
    out = zo.evaluate_h(pack, dom.extended_n, threads)
    assert zo.fr_arr_to_ints(out) == expect


def test_kzg_setup_consistency(oracle):
    """commit_lagrange(evals) == commit(coeffs) == [p(s)]G  (ParamsKZG::setup structure)."""
    zo = oracle
    k = 6
    s = zo.fr_from_int(0x1234567_89ABCDEF)
    mono, lag = zo.kzg_setup_scalars(k, s)
    g, gl = zo.fixed_base_mul(mono, 4), zo.fixed_base_mul(lag, 4)
    dom = zo.Domain(3, k)
    evals = zo.synth_raw253(7, 1 << k)
    coeffs = dom.lagrange_to_coeff(evals)
    c1 = zo.g1_to_affine(zo.best_multiexp(coeffs, g, 2))
    c2 = zo.g1_to_affine(zo.best_multiexp(evals, gl, 2))
    c3 = zo.g1_mul_gen(zo.eval_polynomial(coeffs, s))
    assert (c1 == c2).all() and (c1 == c3).all()


def test_grand_products_and_batch_invert(oracle):
    zo = oracle
    g = load("products.json")
    k, bf = g["k"], g["bf"]
    F = lambda xs: zo.fr_arr_from_ints([H(x) for x in xs])
    values = [F(c) for c in g["values"]]
    sigmas = [F(c) for c in g["sigmas"]]
    beta, gamma = zo.fr_from_int(H(g["beta"])), zo.fr_from_int(H(g["gamma"]))
    blind = np.stack([F(b) for b in g["blinding"]])
    zs = zo.permutation_products(k, values, sigmas, g["chunk_len"], beta, gamma, bf, blind)
    for z, exp in zip(zs, g["z"]):
        assert zo.fr_arr_to_ints(z) == [H(x) for x in exp]
    L = g["lookup"]
    lz = zo.lookup_product(k, F(L["cin"]), F(L["ctab"]), F(L["pin"]), F(L["ptab"]), beta, gamma, bf, F(L["blinding"]))
    assert zo.fr_arr_to_ints(lz) == [H(x) for x in L["z"]]
    bi = g["batch_invert"]
    assert zo.fr_arr_to_ints(zo.batch_invert(F(bi["input"]))) == [H(x) for x in bi["output"]]
    ev = g["evals"]
    assert zo.fr_arr_to_ints(zo.eval_polynomials(values, zo.fr_from_int(H(ev["x"])))) == [H(x) for x in ev["values"]]


def test_permute_expression_pair(oracle):
    zo = oracle
    g = load("products.json")
    pm = g["permute"]
    F = lambda xs: zo.fr_arr_from_ints([H(x) for x in xs])
    pin, ptab = zo.permute_expression_pair(g["k"], g["bf"], F(pm["input"]), F(pm["table"]), F(pm["blind_in"]), F(pm["blind_tab"]))
    assert zo.fr_arr_to_ints(pin) == [H(x) for x in pm["permuted_input"]]
    assert zo.fr_arr_to_ints(ptab) == [H(x) for x in pm["permuted_table"]]
    bad = F(pm["input"])
    bad[0] = zo.fr_from_int(12345678901234567890)       # not in the table
    with pytest.raises(ValueError):
        zo.permute_expression_pair(g["k"], g["bf"], bad, F(pm["table"]), F(pm["blind_in"]), F(pm["blind_tab"]))


def _shplonk_case(zo):
    g = load("shplonk.json")
    F = lambda xs: zo.fr_arr_from_ints([H(x) for x in xs])
    polys = {i: F(c) for i, c in enumerate(g["polys"])}
    queries = [(i, H(pt), H(e)) for i, pt, e in g["queries"]]
    ch = {"shplonk_y": H(g["y"]), "shplonk_v": H(g["v"]), "shplonk_u": H(g["u"])}
    return g, F, polys, queries, ch


def test_shplonk_primitives(oracle):
    zo = oracle
    g, F, polys, _, _ = _shplonk_case(zo)
    lc = g["lincomb"]
    out = zo.linear_combination([polys[i] for i in range(6)], F(lc["coeffs"]), F(lc["low"]))
    assert zo.fr_arr_to_ints(out) == [H(x) for x in lc["out"]]
    dv = g["division"]
    assert zo.fr_arr_to_ints(zo.kate_division(F(dv["dividend"]), F(dv["roots"]))) == [H(x) for x in dv["quotient"]]


def test_shplonk_prover_host_logic(oracle):
    """halo2_zkcert_amd.shplonk (the host side of the product path) driven by the oracle backend: rotation-set grouping,
    h(X), h'(X) and both commitments against the golden vectors (whose generator also checked the verifier equation)."""
    import halo2_zkcert_amd.shplonk as sp
    from oracle_backend import OracleBackend

    zo = oracle
    g, F, polys, queries, ch = _shplonk_case(zo)
    b = OracleBackend(2)
    b.setup(g["k"], 3, H(g["s"]))
    written = []
    pr = sp.ProverSHPLONK(b).create_proof(polys, queries, lambda t: ch[t], lambda t, c: written.append(t))
    assert [len(r.points) for r in pr["rotation_sets"]] == g["set_sizes"]
    assert [[c[0] for c in r.commitments] for r in pr["rotation_sets"]] == g["set_members"]
    assert zo.fr_arr_to_ints(pr["h_x"]) == [H(x) for x in g["h"]]
    assert zo.fr_arr_to_ints(pr["l_x"]) == [H(x) for x in g["h_prime"]]
    assert written == ["shplonk_h1", "shplonk_h2"]
    for got, exp in ((pr["h1"], g["h1"]), (pr["h2"], g["h2"])):
        assert zo.affine_to_ints(got[0].reshape(1, 8))[0] == (H(exp[0]), H(exp[1]))


def test_published_eip196_add_and_mul_vectors(oracle):
    """External anchors for G1 arithmetic on points of UNKNOWN discrete logarithm (everything else here multiplies the generator): go-ethereum's
    published bn256Add / bn256ScalarMul vectors "chfast1" (EIP-196 precompiles; tests/golden/eip196_published.json).  The big-int model
    (oracle/pyref.py) and the C oracle's best_multiexp (oracle/zkoracle.c, the restatement of halo2curves' multiexp the reference pins:
    /root/reference/Cargo.lock:1359-1361) both reproduce them."""
    import json
    import os

    import pyref as P

    zo = oracle
    v = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "eip196_published.json")))
    H = lambda s: int(s, 16)
    pt = lambda xy: (H(xy[0]), H(xy[1]))
    a, b, c = pt(v["add"]["a"]), pt(v["add"]["b"]), pt(v["add"]["sum"])
    assert P.to_affine(P.jac_add(P.from_affine(a), P.from_affine(b))) == c
    m, s, r = pt(v["mul"]["point"]), H(v["mul"]["scalar"]), pt(v["mul"]["product"])
    assert P.to_affine(P.scalar_mul(s, P.from_affine(m))) == r
    one = [1, 1]
    got = zo.affine_to_ints(zo.g1_to_affine(zo.best_multiexp(zo.fr_arr_from_ints(one), zo.affine_from_ints([a, b]), 2)).reshape(1, 8))[0]
    assert got == c
    got = zo.affine_to_ints(zo.g1_to_affine(zo.best_multiexp(zo.fr_arr_from_ints([s, 0]), zo.affine_from_ints([m, a]), 2)).reshape(1, 8))[0]
    assert got == r
    a2, b2, c2 = pt(v["add2"]["a"]), pt(v["add2"]["b"]), pt(v["add2"]["sum"])
    assert P.to_affine(P.jac_add(P.from_affine(a2), P.from_affine(b2))) == c2 == m
    for case in v["mul_more"]:
        pnt, sc, want = pt(case["point"]), H(case["scalar"]) % P.R, pt(case["product"])
        assert P.to_affine(P.scalar_mul(sc, P.from_affine(pnt))) == want
        got = zo.affine_to_ints(zo.g1_to_affine(zo.best_multiexp(zo.fr_arr_from_ints([sc, 1, P.R - 1]), zo.affine_from_ints([pnt, a, a]), 2)).reshape(1, 8))[0]
        assert got == want          # (+ a - a: a three-term sum whose other terms cancel)
