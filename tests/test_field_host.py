"""CPU: the product's own field / curve code (csrc/bn254.hpp is __host__ __device__) through the host-side
self-test entry points of libzkhip.so, against the oracle.  Same source the kernels compile."""
import ctypes as C
import os

import numpy as np
import pytest

R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
P = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47


@pytest.fixture(scope="module")
def zt():
    path = os.environ.get("ZKT_LIB")
    if not path:
        import __graft_entry__ as g

        g.build()
        import halo2_zkcert_amd.ffi as ffi

        path = ffi.LIB_PATH
    return C.CDLL(path)


def edge_values(m, zo, seed):
    vals = [0, 1, 2, 31, 32, 33, m - 1, m - 2, m - 32, (m - 1) // 2, (1 << 253) - 1, (1 << 232), (1 << 232) - 1, (1 << 29) - 1, 1 << 29,
            (1 << 64) - 1, 1 << 128]
    vals += [zo.limbs_to_int(r) % m for r in zo.synth_raw253(seed, 40)]
    return vals


@pytest.mark.parametrize("field", ["fr", "fq"])
def test_mul_inv_chain(zt, oracle, field):
    zo = oracle
    m = R if field == "fr" else P
    frm = getattr(zo, f"{field}_from_int")
    to = getattr(zo, f"{field}_to_int")
    vals = edge_values(m, zo, 900)
    o = zo.new(4)
    for i, a in enumerate(vals):
        b = vals[(i * 5 + 3) % len(vals)]
        A, B = frm(a), frm(b)
        getattr(zt, f"zkt_{field}_mul")(zo.p(A), zo.p(B), zo.p(o))
        assert to(o) == a * b % m
        getattr(zt, f"zkt_{field}_chain")(zo.p(A), zo.p(B), zo.p(o))
        mm = (a * a - b * b) % m
        t = (2 * mm + a + 8) % m
        assert to(o) == (t * t - mm) % m
        getattr(zt, f"zkt_{field}_inv")(zo.p(A), zo.p(o))
        assert to(o) == (pow(a, -1, m) if a else 0)
        getattr(zt, f"zkt_{field}_inv_host")(zo.p(A), zo.p(o))      # binary extended Euclid used on the host paths
        assert to(o) == (pow(a, -1, m) if a else 0)
        getattr(zt, f"zkt_{field}_inv_euclid")(zo.p(A), zo.p(o))    # the 8 x u32 version the batch-inversion kernel runs
        assert to(o) == (pow(a, -1, m) if a else 0)


def test_conversions(zt, oracle):
    zo = oracle
    o = zo.new(4)
    for a in edge_values(R, zo, 901):
        A = zo.fr_from_int(a)
        zt.zkt_fr_raw_roundtrip(zo.p(A), zo.p(o))
        assert (o == A).all()
        zt.zkt_fr_x32_roundtrip(zo.p(A), zo.p(o))
        assert (o == A).all()
        zt.zkt_fr_to_canonical(zo.p(A), zo.p(o))
        assert zo.limbs_to_int(o) == a
        c = zo.int_to_limbs(a)
        zt.zkt_fr_from_canonical(zo.p(c), zo.p(o))
        assert (o == A).all()


def test_canonical_of_every_multiple(zt, oracle):
    zo = oracle
    for field, m in ((1, R), (0, P)):
        for a in (0, 1, m - 1, m // 2, (1 << 232) - 1, 1 << 232):
            raw = zo.int_to_limbs(a)
            for k in range(0, 121):
                assert zt.zkt_canon_kp(zo.p(raw), C.c_uint32(k), C.c_int(field)) == 1, (field, hex(a), k)


def test_g1_ops(zt, oracle):
    zo = oracle
    from util import H, load

    g = load("g1.json")
    ops = zo.affine_from_ints([(H(o["x"]), H(o["y"])) for o in g["add_operands"]])
    jac = []
    for a in ops:
        j = zo.new(12)
        zo.lib().zko_g1_from_affine(zo.p(a), zo.p(j))
        # non-trivial representative: scale by z
        z = zo.fq_from_int(0xABCDEF12345)
        z2 = zo._binary("zko_fq_mul", z, z)
        z3 = zo._binary("zko_fq_mul", z2, z)
        if (j[8:] != 0).any():
            j = np.concatenate([zo._binary("zko_fq_mul", j[:4], z2), zo._binary("zko_fq_mul", j[4:8], z3), z])
        jac.append(j)
    import halo2_zkcert_amd.ffi as ffi

    for c in g["adds"]:
        o = zo.new(12)
        zt.zkhip_g1_add(zo.p(jac[c["i"]]), zo.p(jac[c["j"]]), zo.p(o))
        assert zo.affine_to_ints(zo.g1_to_affine(o))[0] == (H(c["x"]), H(c["y"]))
        a8 = zo.new(8)
        zt.zkhip_g1_to_affine(zo.p(o), zo.p(a8))
        assert zo.affine_to_ints(a8)[0] == (H(c["x"]), H(c["y"]))
        zt.zkt_g1_add_mixed(zo.p(jac[c["i"]]), zo.p(ops[c["j"]]), zo.p(o))
        assert zo.affine_to_ints(zo.g1_to_affine(o))[0] == (H(c["x"]), H(c["y"]))
    for j in jac:
        o, e = zo.new(12), zo.new(12)
        zt.zkt_g1_double(zo.p(j), zo.p(o))
        zo.lib().zko_g1_double(zo.p(j), zo.p(e))
        assert (zo.g1_to_affine(o) == zo.g1_to_affine(e)).all()


def test_long_mixed_sum_keeps_invariants(zt, oracle):
    """2000 mixed additions with random signs (and repeated points -> doubling / cancellation cases):
    the lazy-bound loop invariant of g1j_add_mixed holds in practice, not just in the types."""
    zo = oracle
    n = 2000
    pts = zo.fixed_base_mul(zo.fr_arr_from_ints([zo.limbs_to_int(r) % R for r in zo.synth_raw253(77, n)]), 8)
    pts[5] = pts[4]            # P + P
    pts[9] = pts[8]            # P - P (with signs below)
    pts[20] = 0                # identity point in the table
    negs = np.random.default_rng(1).integers(0, 2, n).astype(np.uint8)
    negs[4] = negs[5] = 0
    negs[8], negs[9] = 0, 1
    # the XYZZ sum alternates between two accumulators (even / odd index): make the exceptional cases of g1x_add_mixed happen INSIDE one
    # accumulator — acc == q exactly (index 0 then 2: the doubling branch of the fix-up), acc == -q (index 1 then 3: the sum becomes the
    # identity, and index 5 then meets an identity accumulator) — the formula-first / unlikely-fix-up form of round 3 must take them
    pts[2] = pts[0]
    negs[0] = negs[2] = 0
    pts[3] = pts[1]
    negs[1], negs[3] = 0, 1
    o = zo.new(12)
    zt.zkt_g1_sum_mixed(zo.p(pts), negs.ctypes.data_as(C.c_void_p), C.c_size_t(n), zo.p(o))
    sc = zo.fr_arr_from_ints([(R - 1) if s else 1 for s in negs])
    exp = zo.g1_to_affine(zo.best_multiexp(sc, pts, 8))
    assert (zo.g1_to_affine(o) == exp).all()
    # XYZZ accumulators (mixed add, full add, double, storage round trip): 3 * the same sum
    zt.zkt_g1x_sum_mixed(zo.p(pts), negs.ctypes.data_as(C.c_void_p), C.c_size_t(n), zo.p(o))
    sc3 = zo.fr_arr_from_ints([(R - 3) if s else 3 for s in negs])
    exp3 = zo.g1_to_affine(zo.best_multiexp(sc3, pts, 8))
    assert (zo.g1_to_affine(o) == exp3).all()


def test_batch_to_affine(zt, oracle):
    zo = oracle
    import halo2_zkcert_amd.ffi as ffi

    n = 9
    pts = zo.fixed_base_mul(zo.fr_arr_from_ints(list(range(3, 3 + n))), 2)
    z = zo.fq_from_int(0x9876543210F)
    z2 = zo._binary("zko_fq_mul", z, z)
    z3 = zo._binary("zko_fq_mul", z2, z)
    jac = np.zeros((n, 12), dtype=np.uint64)
    for i in range(n):
        jac[i] = np.concatenate([zo._binary("zko_fq_mul", pts[i, :4], z2), zo._binary("zko_fq_mul", pts[i, 4:], z3), z])
    jac[4] = 0                      # identity in the middle of the batch
    exp = pts.copy()
    exp[4] = 0
    assert (ffi.g1_batch_to_affine(jac) == exp).all()
