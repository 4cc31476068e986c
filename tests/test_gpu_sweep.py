"""GPU parity: the quotient sweep (evaluate_h) vs the golden vector and the CPU oracle."""
import numpy as np
import pytest

from util import H, evalh_case, load

pytestmark = pytest.mark.gpu


def _to_device_kw(ctx, kw):
    kw = dict(kw)
    for key in ("fixed", "advice", "instance", "sigma", "perm_z", "lookup_z", "lookup_a", "lookup_s"):
        kw[key] = [ctx.to_device(c) for c in kw[key]]
    for key in ("l0", "l_last", "l_active"):
        kw[key] = ctx.to_device(kw[key])
    return kw


def test_evaluate_h_golden(zk, oracle):
    ffi, ctx = zk
    zo = oracle
    dom, kw, expect = evalh_case(zo)
    pack = ffi.EvalhPack()
    pack.build(**_to_device_kw(ctx, kw))
    out = ctx.to_host(ffi.evaluate_h(ctx, pack, dom.extended_n))
    assert zo.fr_arr_to_ints(out) == expect


def random_circuit(zo, k, degree, bf, n_adv, n_fix, n_lookups, n_perm, seed):
    """Random gate set in the shape of halo2-lib's BaseConfig plus extras; random (non-satisfying) columns."""
    import halo2_zkcert_amd.evaluator as ev

    rng = np.random.default_rng(seed)
    dom = zo.Domain(degree, k)
    A = lambda c, r: ("advice", c, r)
    F = lambda c, r: ("fixed", c, r)
    gates = []
    for c in range(n_adv):
        gates.append(("prod", F(c % n_fix, 0), ("sum", ("sum", A(c, 0), ("prod", A(c, 1), A(c, 2))), ("neg", A(c, 3)))))
    gates.append(("sum", ("prod", A(0, 0), A(0, 0)), ("neg", ("scaled", ("instance", 0, -1), 12345))))
    gates.append(("prod", ("sum", ("challenge", 0), A(n_adv - 1, -2)), ("sum", F(0, 1), ("const", 7))))
    lookups = [ev.build_lookup([A(i % n_adv, 0), ("prod", F(0, 0), A(0, 1))][: 1 + i % 2], [F(n_fix - 1, 0), F(0, 0)][: 1 + i % 2])
               for i in range(n_lookups)]
    cols = {}
    s = [seed * 100]

    def col():
        s[0] += 1
        return dom.coeff_to_extended(zo.synth_raw253(s[0], 1 << k), 8)

    perm_columns = ([("advice", i) for i in range(n_adv)] + [("fixed", 0), ("instance", 0)])[:n_perm]
    n_sets = -(-len(perm_columns) // (degree - 2))
    l0, ll, la = dom.l_cosets(bf, 8)
    zeta, delta = zo.fr_constants()
    ch = zo.synth_raw253(seed + 9, 5)
    kw = dict(k=k, extended_k=dom.extended_k, cs_degree=degree, blinding_factors=bf, extended_omega=dom.extended_omega,
              g_coset=dom.g_coset, delta=delta, beta=ch[0], gamma=ch[1], theta=ch[2], y=ch[3],
              fixed=[col() for _ in range(n_fix)], advice=[col() for _ in range(n_adv)], instance=[col()],
              challenges=ch[4:5], l0=l0, l_last=ll, l_active=la, gates_graph=ev.build_custom_gates(gates),
              perm_columns=perm_columns, sigma=[col() for _ in perm_columns], perm_z=[col() for _ in range(n_sets)],
              lookup_graphs=lookups, lookup_z=[col() for _ in lookups], lookup_a=[col() for _ in lookups],
              lookup_s=[col() for _ in lookups], to_mont=lambda xs: zo.fr_arr_from_ints(xs))
    return dom, kw


@pytest.mark.parametrize("cfg", [
    dict(k=8, degree=4, bf=6, n_adv=4, n_fix=3, n_lookups=1, n_perm=6, seed=1),     # RSA-shaped
    dict(k=7, degree=5, bf=5, n_adv=6, n_fix=2, n_lookups=0, n_perm=0, seed=2),     # SHA-shaped: no lookup, no permutation
    dict(k=6, degree=3, bf=3, n_adv=2, n_fix=2, n_lookups=2, n_perm=3, seed=3),     # chunk_len 1, two lookups
    dict(k=9, degree=9, bf=4, n_adv=3, n_fix=1, n_lookups=1, n_perm=5, seed=4),     # extension factor 8
])
def test_evaluate_h_vs_oracle(zk, oracle, cfg):
    ffi, ctx = zk
    zo = oracle
    dom, kw = random_circuit(zo, **cfg)
    opack = zo.EvalhPack()
    opack.build(**kw)
    exp = zo.evaluate_h(opack, dom.extended_n, 8)
    pack = ffi.EvalhPack()
    pack.build(**_to_device_kw(ctx, kw))
    got = ctx.to_host(ffi.evaluate_h(ctx, pack, dom.extended_n))
    assert (got == exp).all()


def test_evaluate_h_row_ranges(zk, oracle):
    """zkhip_evaluate_h_rows_device (one rank's share of a row-sharded sweep): uneven 64-multiples of the extended rows, glued
    together, are the full sweep — rotations read across the range boundaries and wrap around the domain; bad ranges are refused."""
    ffi, ctx = zk
    zo = oracle
    dom, kw = random_circuit(zo, k=8, degree=4, bf=6, n_adv=4, n_fix=3, n_lookups=1, n_perm=6, seed=11)
    opack = zo.EvalhPack()
    opack.build(**kw)
    exp = zo.evaluate_h(opack, dom.extended_n, 8)
    pack = ffi.EvalhPack()
    pack.build(**_to_device_kw(ctx, kw))
    en = dom.extended_n
    cuts = [0, 64, 64 + 128, en // 2, en - 192, en]
    parts = [ctx.to_host(ffi.evaluate_h_rows(ctx, pack, a, b - a)) for a, b in zip(cuts, cuts[1:])]
    assert (np.concatenate(parts) == exp).all()
    for first, cnt in ((0, 100), (en - 64, 128), (0, 0)):
        with pytest.raises(ffi.ZkhipError):
            ffi.evaluate_h_rows(ctx, pack, first, cnt)


def test_evaluate_h_rejects_malformed_program(zk, oracle):
    ffi, ctx = zk
    zo = oracle
    dom, kw = random_circuit(zo, k=6, degree=3, bf=3, n_adv=2, n_fix=2, n_lookups=0, n_perm=0, seed=5)
    pack = ffi.EvalhPack()
    pack.build(**_to_device_kw(ctx, kw))
    pack.args.custom_gates.n_code_words -= 2      # truncated stream
    with pytest.raises(ffi.ZkhipError):
        ffi.evaluate_h(ctx, pack, dom.extended_n)
