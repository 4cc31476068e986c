"""GPU parity: MSM / KZG commitments through the C ABI vs the golden vectors and the CPU oracle."""
import numpy as np
import pytest

from util import H, load

pytestmark = pytest.mark.gpu


def aff(zk, jac):
    ffi, _ = zk
    return ffi.g1_to_affine(jac)


def test_golden_cases(zk, oracle):
    ffi, ctx = zk
    zo = oracle
    for c in load("msm.json")["cases"]:
        sc = zo.fr_arr_from_ints([H(s) for s in c["scalars"]])
        pts = zo.affine_from_ints([(H(x), H(y)) for x, y in c["points"]])
        res = aff(zk, ffi.best_multiexp(ctx, sc, pts))
        assert zo.affine_to_ints(res)[0] == (H(c["result"][0]), H(c["result"][1])), c["name"]


def test_golden_seeded_1024(zk, oracle):
    ffi, ctx = zk
    zo = oracle
    s = load("msm.json")["seeded"]
    sc = zo.synth_raw253(s["seed_scalars"], s["n"])
    pts = zo.fixed_base_mul(zo.fr_arr_from_ints(zo.arr_to_ints(zo.synth_raw253(s["seed_points"], s["n"]))), threads=8)
    res = aff(zk, ffi.best_multiexp(ctx, sc, pts))
    assert zo.affine_to_ints(res)[0] == (H(s["result"][0]), H(s["result"][1]))
    # the device generator agrees with the spec'd PRNG
    t = ctx.synth_fill(s["n"], s["seed_scalars"])
    assert (ctx.to_host(t) == sc).all()


@pytest.fixture(scope="module")
def params12(zk, oracle):
    ffi, ctx = zk
    s = oracle.fr_from_int(0xC0FFEE1234567)
    p = ffi.ParamsKZG.setup(ctx, 12, s)
    yield p, s
    p.free()


def test_setup_matches_oracle(zk, oracle, params12):
    ffi, ctx = zk
    zo = oracle
    p, s = params12
    mono, lag = zo.kzg_setup_scalars(12, s)
    idx = [0, 1, 2, 77, 4095]
    g = p.read_bases(p.g, 0, 4096)
    gl = p.read_bases(p.g_lagrange, 0, 4096)
    exp_g = zo.fixed_base_mul(mono[idx], 2)
    exp_gl = zo.fixed_base_mul(lag[idx], 2)
    assert (g[idx] == exp_g).all() and (gl[idx] == exp_gl).all()


@pytest.mark.parametrize("n", [1, 2, 63, 1000, 4096])
def test_commit_vs_oracle_ragged(zk, oracle, params12, n):
    ffi, ctx = zk
    zo = oracle
    p, _ = params12
    bases = p.read_bases(p.g, 0, n)
    sc = zo.synth_raw253(1000 + n, n)
    exp = zo.g1_to_affine(zo.best_multiexp(sc, bases, 8))
    got = aff(zk, p.commit(sc))
    assert (got == exp).all()


def test_commit_empty_and_zero(zk, params12):
    ffi, ctx = zk
    p, _ = params12
    assert (aff(zk, p.commit(np.zeros((0, 4), dtype=np.uint64))) == 0).all()
    assert (aff(zk, p.commit(np.zeros((300, 4), dtype=np.uint64))) == 0).all()
    with pytest.raises(ffi.ZkhipError):
        p.commit(np.zeros((4097, 4), dtype=np.uint64))     # more scalars than bases


@pytest.mark.parametrize("kind", ["bool", "bytes", "same", "sparse", "rminus1"])
def test_commit_skewed_scalars(zk, oracle, params12, kind):
    """Witness-like distributions: every scalar in one bucket, tiny values, mostly zero."""
    ffi, ctx = zk
    zo = oracle
    p, _ = params12
    n = 4096
    rng = np.random.default_rng(7)
    if kind == "bool":
        vals = [int(v) for v in rng.integers(0, 2, n)]
    elif kind == "bytes":
        vals = [int(v) for v in rng.integers(0, 256, n)]
    elif kind == "same":
        vals = [0x1234567] * n
    elif kind == "sparse":
        vals = [0] * n
        for i in range(0, n, 97):
            vals[i] = int(rng.integers(1, 1 << 62))
    else:
        R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
        vals = [R - 1 - (i % 3) for i in range(n)]
    sc = zo.fr_arr_from_ints(vals)
    bases = p.read_bases(p.g, 0, n)
    exp = zo.g1_to_affine(zo.best_multiexp(sc, bases, 8))
    assert (aff(zk, p.commit(sc)) == exp).all()


def test_synth_small_columns_and_their_commitments(zk, oracle, params12):
    """zkhip_synth_small_device (SURVEY 8(d)'s bit / word columns) against its numpy restatement, and a batch of such columns —
    the sparse-digit path: device-adaptive lane width, one heavy bucket — against the oracle's MSM."""
    from oracle_backend import OracleBackend

    ffi, ctx = zk
    zo = oracle
    p, _ = params12
    n = 4096
    ob = OracleBackend()
    cols_d = [ctx.synth_small(n, 900 + j, bpm, wb) for j, (bpm, wb) in enumerate([(900, 32), (1000, 1), (0, 16), (500, 8)])]
    cols_h = [ob.synth_small(n, 900 + j, bpm, wb) for j, (bpm, wb) in enumerate([(900, 32), (1000, 1), (0, 16), (500, 8)])]
    for d, h in zip(cols_d, cols_h):
        assert (ctx.to_host(d) == h).all()
    ints = zo.fr_arr_to_ints(cols_h[0])
    assert max(ints) < 1 << 32 and 0.85 < sum(1 for v in ints if v < 2) / n < 0.95
    out = ctx.to_host(p.commit_batch_device(cols_d, lagrange=True))
    bases = p.read_bases(p.g_lagrange, 0, n)
    for j, c in enumerate(cols_h):
        assert (ffi.g1_to_affine(out[j]) == zo.g1_to_affine(zo.best_multiexp(c, bases, 8))).all()


def test_commit_batch_device(zk, oracle, params12):
    ffi, ctx = zk
    zo = oracle
    p, _ = params12
    n = 4096
    cols_h = [zo.synth_raw253(500 + j, n) for j in range(3)]
    cols_h.append(np.zeros((n, 4), dtype=np.uint64))
    cols = [ctx.to_device(c) for c in cols_h]
    out = ctx.to_host(p.commit_batch_device(cols, lagrange=True))
    bases = p.read_bases(p.g_lagrange, 0, n)
    for j, c in enumerate(cols_h):
        exp = zo.g1_to_affine(zo.best_multiexp(c, bases, 8))
        assert (ffi.g1_to_affine(out[j]) == exp).all()


@pytest.mark.parametrize("tail2", [0, 1])
def test_commit_batch_both_tail_forms(zk, oracle, params12, tail2):
    """The tail's two forms — the chunk's base applied per chunk, or to the chunk totals by a second running-sum pass (k_chunk_totals;
    the default only for wide batches at >= 4096 chunks) — forced in turn on a batch with a zero column, a skewed one and ragged lengths."""
    ffi, ctx = zk
    zo = oracle
    p, _ = params12
    ctx.set_option("msm_tail2", tail2)
    try:
        for n in (4096, 1000):
            cols_h = [zo.synth_raw253(900 + j + n, n) for j in range(2)]
            cols_h.append(np.zeros((n, 4), dtype=np.uint64))
            ones = np.zeros((n, 4), dtype=np.uint64)
            ones[:, 0] = 1                                    # every scalar the same small Montgomery-form value: one bucket takes a whole window
            cols_h.append(ones)
            out = ctx.to_host(p.commit_batch_device([ctx.to_device(c) for c in cols_h], lagrange=True))
            bases = p.read_bases(p.g_lagrange, 0, n)
            for j, c in enumerate(cols_h):
                assert (ffi.g1_to_affine(out[j]) == zo.g1_to_affine(zo.best_multiexp(c, bases, 8))).all(), (tail2, n, j)
    finally:
        ctx.set_option("msm_tail2", -1)


def test_commit_lagrange_equals_commit_coeff_equals_trapdoor(zk, oracle, params12):
    """commit_lagrange(evals) == commit(iNTT(evals)) == [p(s)] G — ties MSM, NTT and the SRS together."""
    ffi, ctx = zk
    zo = oracle
    p, s = params12
    dom = ffi.EvaluationDomain(ctx, 3, 12)
    evals = zo.synth_raw253(77, 4096)
    coeffs = dom.lagrange_to_coeff(evals)
    c1 = aff(zk, p.commit(coeffs))
    c2 = aff(zk, p.commit_lagrange(evals))
    c3 = zo.g1_mul_gen(zo.eval_polynomial(coeffs, s))
    assert (c1 == c2).all() and (c1 == c3).all()
    assert ffi.g1_to_bytes(c1) == zo.g1_to_bytes(c3)
    dom.free()


@pytest.mark.parametrize("k", [17])
def test_full_size_trapdoor_property(zk, oracle, k):
    """BASELINE config size (2^17 points): MSM(coeffs, SRS) == [p(s)] G with p(s) by Horner on the host."""
    ffi, ctx = zk
    zo = oracle
    s = zo.fr_from_int(0xDEADBEEF12345)
    p = ffi.ParamsKZG.setup(ctx, k, s)
    n = 1 << k
    col = ctx.synth_fill(n, 0xC0FFEE17)
    out = ctx.to_host(p.commit_batch_device([col]))
    coeffs = ctx.to_host(col)
    exp = zo.g1_mul_gen(zo.eval_polynomial(coeffs, s))
    assert (ffi.g1_to_affine(out[0]) == exp).all()
    p.free()


@pytest.mark.parametrize("k", [20, 22] + ([24] if __import__("os").environ.get("ZKHIP_TEST_HUGE") else []))   # 2^24: ~1 min, 20 GB
def test_large_msm_trapdoor_property(zk, oracle, k):
    """BASELINE configs[2..3] sizes (2^20, 2^22 points; window c = 18): commit(coeffs) == [p(s)] G,
    commit_lagrange(evals) == commit(iNTT(evals)).  Size-independent, O(n) host work."""
    ffi, ctx = zk
    zo = oracle
    s = zo.fr_from_int(0xFEEDFACE0000 + k)
    p = ffi.ParamsKZG.setup(ctx, k, s)
    n = 1 << k
    col = ctx.synth_fill(n, 0xC0FFEE00 + k)
    coeffs = ctx.to_host(col)
    out = ctx.to_host(p.commit_batch_device([col]))
    exp = zo.g1_mul_gen(zo.eval_polynomial(coeffs, s))
    assert (ffi.g1_to_affine(out[0]) == exp).all()
    dom = ffi.EvaluationDomain(ctx, 3, k)
    ev = col.clone()
    dom.coeff_to_lagrange_device([ev])
    out2 = ctx.to_host(p.commit_batch_device([ev], lagrange=True))
    assert (ffi.g1_to_affine(out2[0]) == exp).all()
    dom.free()
    p.free()


def test_pipelined_host_slice_msm_equals_the_device_resident_sum(zk, oracle):
    """zkhip_msm_g1 on a caller's HOST slice (best_multiexp at the `curves` patch level: /root/reference/src/helpers.rs:233,299, src/bin/cli.rs:320,369,519
    through ParamsKZG::commit): from 2^20 scalars the call is pipelined — the slice uploaded in K chunks, chunk j's whole MSM over points
    [off_j, off_j + len_j) started when its bytes have landed, on two streams alternately, the K partial sums added on the host.  Same point as
    the device-resident one-column MSM and as the trapdoor value, for every K (option msm_host_chunks), for ragged lengths, from pageable and from
    pinned memory; an error inside restores the context's stream."""
    import torch

    ffi, ctx = zk
    zo = oracle
    k = 21
    n = 1 << k
    s = zo.fr_from_int(0xFEEDFACE7777)
    p = ffi.ParamsKZG.setup(ctx, k, s)
    d_col = ctx.synth_fill(n, 0xC0FFEE21)
    host = ctx.to_host(d_col).copy()                                   # numpy-owned: pageable
    pin = torch.empty((n, 4), dtype=torch.int64).pin_memory()
    pin.numpy().view(np.uint64)[:] = host
    want = ffi.g1_to_affine(ctx.to_host(p.commit_batch_device([d_col]))[0])
    assert (want == zo.g1_mul_gen(zo.eval_polynomial(host, s))).all()
    try:
        for chunks in (0, 1, 2, 3, 4, 8):                              # 0: by size (4 at 2^21)
            ctx.set_option("msm_host_chunks", chunks)
            assert (ffi.g1_to_affine(p.commit(host)) == want).all(), chunks
        assert (ffi.g1_to_affine(p.commit(pin.numpy().view(np.uint64))) == want).all()
        # ragged: the last chunk is short; a length that is not a multiple of the sort's 256-scalar blocks; chunk boundaries move with n
        for m in (n - 1, n - 255, (1 << 20) + 12345, (3 << 19) + 1):
            ref = ffi.g1_to_affine(ctx.to_host(p.commit_batch_device([d_col], n=m))[0])
            for chunks in (0, 3, 4):
                ctx.set_option("msm_host_chunks", chunks)
                assert (ffi.g1_to_affine(p.commit(host[:m])) == ref).all(), (m, chunks)
        # all-zero and single-non-zero slices through the pipelined form
        ctx.set_option("msm_host_chunks", 4)
        z = np.zeros((n, 4), dtype=np.uint64)
        assert (ffi.g1_to_affine(p.commit(z)) == 0).all()
        z[n - 7] = zo.fr_from_int(5)
        bases = p.read_bases(p.g, n - 7, 1)
        assert (ffi.g1_to_affine(p.commit(z)) == zo.g1_to_affine(zo.best_multiexp(z[n - 7:n - 6], bases, 1))).all()
    finally:
        ctx.set_option("msm_host_chunks", 0)
    # back to back with device-resident work on the same context: the pipelined call leaves the context on its main stream
    assert (ffi.g1_to_affine(ctx.to_host(p.commit_batch_device([d_col]))[0]) == want).all()
    p.free()


@pytest.mark.parametrize("k", [16, 20])
def test_host_column_batch_equals_the_device_batch(zk, oracle, k):
    """zkhip_msm_g1_batch (several HOST columns in one call — what a patched ParamsKZG would hand over for a batch of commitments, the callers being
    /root/reference/src/helpers.rs:233,299 through create_proof's commitment loops): the first column through the chunk pipeline, the others in device batches of up to
    four as their uploads land.  Every sum equals the device-resident batch's, for 1 / 2 / 5 / 6 columns, ragged lengths, a repeated column, an all-zero column, over the
    Lagrange basis too; below 2^16 scalars the call is a loop over zkhip_msm_g1."""
    ffi, ctx = zk
    zo = oracle
    n = 1 << k
    p = ffi.ParamsKZG.setup(ctx, k, zo.fr_from_int(0xBA7C4000 + k))
    d_cols = [ctx.synth_fill(n, 0xBA7C0 + j) for j in range(6)]
    host = [ctx.to_host(c).copy() for c in d_cols]
    host[3][:] = 0                                       # an all-zero column in the middle of a batch
    d_cols[3] = ctx.to_device(host[3])
    want = ffi.g1_batch_to_affine(ctx.to_host(p.commit_batch_device(d_cols)))
    for ncols in (1, 2, 5, 6):
        got = p.commit_batch_host(host[:ncols])
        assert (ffi.g1_batch_to_affine(got) == want[:ncols]).all(), ncols
    assert (ffi.g1_batch_to_affine(p.commit_batch_host([host[2], host[2], host[0]])) == want[[2, 2, 0]]).all()      # the same host array twice
    for m in (n - 1, n - 300, n // 2 + 77):
        ref = ffi.g1_batch_to_affine(ctx.to_host(p.commit_batch_device(d_cols[:3], n=m)))
        assert (ffi.g1_batch_to_affine(p.commit_batch_host([h[:m] for h in host[:3]])) == ref).all(), m
    lag = ffi.g1_batch_to_affine(ctx.to_host(p.commit_batch_device(d_cols[:2], lagrange=True)))
    assert (ffi.g1_batch_to_affine(p.commit_batch_host(host[:2], lagrange=True)) == lag).all()
    assert p.commit_batch_host([]).shape == (0, 12)
    small = [h[:1000] for h in host[:3]]                 # below the pipeline's floor: a loop over the plain form
    ref = ffi.g1_batch_to_affine(ctx.to_host(p.commit_batch_device(d_cols[:3], n=1000)))
    assert (ffi.g1_batch_to_affine(p.commit_batch_host(small)) == ref).all()
    # and the context is back on its main stream: device work right behind it
    assert (ffi.g1_batch_to_affine(ctx.to_host(p.commit_batch_device(d_cols))) == want).all()
    p.free()


@pytest.mark.parametrize("k", [18, 19])
def test_c17_window_path(zk, oracle, k):
    """2^18 and 2^19 points: the only sizes that select the window c = 17 / W = 15 (15 x 17 = 255 bits, no short top window) —
    BASELINE configs[2]'s MSM size.  (a) commit(coeffs) == [p(s)] G and commit_lagrange(evals) == commit(iNTT(evals)) (size-independent
    trapdoor property); (b) a batch of columns with witness-shaped values (bits, 32-bit words, all-equal, zero) against the same
    identity; (c) the first 4096 points against the CPU oracle's best_multiexp, compared as canonical affine bytes."""
    ffi, ctx = zk
    zo = oracle
    s = zo.fr_from_int(0xC17C17000 + k)
    p = ffi.ParamsKZG.setup(ctx, k, s)
    assert p.window() == (17, 15)
    n = 1 << k
    col = ctx.synth_fill(n, 0xC0FFEE00 + k)
    coeffs = ctx.to_host(col)
    exp = zo.g1_mul_gen(zo.eval_polynomial(coeffs, s))
    out = ctx.to_host(p.commit_batch_device([col]))
    assert (ffi.g1_to_affine(out[0]) == exp).all()
    dom = ffi.EvaluationDomain(ctx, 3, k)
    ev = col.clone()
    dom.coeff_to_lagrange_device([ev])
    out2 = ctx.to_host(p.commit_batch_device([ev], lagrange=True))
    assert (ffi.g1_to_affine(out2[0]) == exp).all()
    dom.free()
    # (b) skewed columns in one batch
    import torch
    bits = ctx.synth_small(n, 5, 1000, 1)
    words = ctx.synth_small(n, 6, 0, 32)
    same = col[:1].expand(n, 4).contiguous()
    zero = torch.zeros_like(col)
    batch = [bits, words, same, zero, col]
    outs = ctx.to_host(p.commit_batch_device(batch))
    for c_, o in zip(batch, outs):
        e = zo.g1_mul_gen(zo.eval_polynomial(ctx.to_host(c_), s))
        assert ffi.g1_to_bytes(ffi.g1_to_affine(o)) == zo.g1_to_bytes(e)
    # (c) a point range against the oracle's Pippenger on the same bases
    m = 4096
    bases = p.read_bases(p.g, 0, m)
    part = ctx.to_host(p.commit_batch_device([col], n=m, first=0))
    ref = zo.g1_to_affine(zo.best_multiexp(coeffs[:m], bases, 8))
    assert (ffi.g1_to_affine(part[0]) == ref).all()
    p.free()


def test_quad_cooperative_point_ops(zk, oracle):
    """The 4-lane cooperative XYZZ addition / doubling of the MSM tail against the one-lane versions, including P + P,
    P + (-P), identity operands and non-trivial ZZ / ZZZ."""
    import ctypes as C

    ffi, ctx = zk
    zo = oracle
    rng = np.random.default_rng(5)
    P_MOD = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
    R261 = pow(2, 261, P_MOD)

    def xyzz(k, z):   # k*G in XYZZ coordinates with Z = z, raw R' (2^261) form limbs
        if k == 0:
            return [0, R261, 0, 0]
        x, y = zo.affine_to_ints(zo.g1_mul_gen(zo.fr_from_int(k)).reshape(1, 8))[0]
        zz, zzz = z * z % P_MOD, z * z * z % P_MOD
        return [v * R261 % P_MOD for v in (x * zz % P_MOD, y * zzz % P_MOD, zz, zzz)]

    cases = [(3, 1, 5, 1), (3, 7, 3, 11), (3, 7, P_MOD and -3, 11), (0, 1, 9, 4), (9, 4, 0, 1), (0, 1, 0, 1)]
    cases += [(int(rng.integers(1, 1 << 40)), int(rng.integers(2, 1 << 60)), int(rng.integers(1, 1 << 40)), int(rng.integers(2, 1 << 60))) for _ in range(58)]
    R_ORDER = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
    flat = []
    for ka, za, kb, zb in cases:
        flat += xyzz(ka % R_ORDER, za) + xyzz(kb % R_ORDER, zb)
    arr = np.array([[(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)] for v in flat], dtype=np.uint64)
    d_in = ctx.to_device(arr)
    d_out = ctx.empty(len(cases) * 16)
    assert ffi.lib().zkt_quad_selfcheck(ctx.h, C.c_void_p(d_in.data_ptr()), C.c_uint32(len(cases)), C.c_void_p(d_out.data_ptr())) == 0
    out = ctx.to_host(d_out).reshape(len(cases), 4, 4, 4)
    for i, c in enumerate(cases):
        assert (out[i, 0] == out[i, 2]).all(), ("add", c)
        assert (out[i, 1] == out[i, 3]).all(), ("double", c, (out[i, 1] == out[i, 3]).all(axis=1))


def test_params_file_round_trip(zk, oracle, tmp_path):
    """ParamsKZG::write -> ::read (RawBytes layout): same file size as upstream's, and the reloaded SRS commits identically."""
    ffi, ctx = zk
    zo = oracle
    k = 10
    p = ffi.ParamsKZG.setup(ctx, k, zo.fr_from_int(0xABCDEF123))
    path = str(tmp_path / f"kzg_bn254_{k}.srs")
    p.write(path)
    import os
    assert os.path.getsize(path) == 4 + 2 * (1 << k) * 64 + 256
    q = ffi.ParamsKZG.read(ctx, path)
    poly = zo.synth_raw253(31, 1 << k)
    assert (aff(zk, p.commit(poly)) == aff(zk, q.commit(poly))).all()
    assert (aff(zk, p.commit_lagrange(poly)) == aff(zk, q.commit_lagrange(poly))).all()
    # first two monomial bases are G and [s]G: the file starts with k then the generator (1, 2) in Montgomery form
    with open(path, "rb") as f:
        head = f.read(4 + 64)
    assert int.from_bytes(head[:4], "little") == k
    g = np.frombuffer(head[4:], dtype="<u8").reshape(1, 8)
    assert zo.affine_to_ints(g)[0] == (1, 2)
    p.free(); q.free()


def test_point_range_shard_handles(zk, oracle):
    """zkhip_kzg_setup_range / zkhip_srs_load_range: a handle that holds the window tables of a point range only.  Without a communicator
    an MSM over GLOBAL indices inside the shard equals the same range over the whole SRS, the shard's bases read back equal the whole
    SRS's, and a range that leaves the shard is an error (with a communicator it would be the collective path, tests/test_gpu_distributed.py)."""
    import ctypes as C

    ffi, ctx = zk
    zo = oracle
    k, n = 12, 1 << 12
    s = zo.fr_from_int(0x5AA5D)
    whole = ffi.ParamsKZG.setup(ctx, k, s)
    first, count = 1000, 1500
    g, gl = C.c_void_p(), C.c_void_p()
    ffi._check(ffi.lib().zkhip_kzg_setup_range(ctx.h, C.c_uint32(k), ffi._p(ffi._u64(s)), C.c_size_t(first), C.c_size_t(count), C.byref(g), C.byref(gl)))
    shard = ffi.ParamsKZG(ctx, k, g, gl)
    assert shard.range() == (first, count, n) and whole.range() == (0, n, n)
    assert (shard.read_bases(shard.g, first + 7, 20) == whole.read_bases(whole.g, first + 7, 20)).all()
    assert (shard.read_bases(shard.g_lagrange, first, 5) == whole.read_bases(whole.g_lagrange, first, 5)).all()
    # the same shard from host bases (a slice of a params file)
    h = C.c_void_p()
    bases = whole.read_bases(whole.g, first, count)
    ffi._check(ffi.lib().zkhip_srs_load_range(ctx.h, ffi._p(bases), C.c_size_t(n), C.c_size_t(first), C.c_size_t(count), C.byref(h)))
    loaded = ffi.ParamsKZG(ctx, k, h, None)
    cols = [ctx.synth_fill(n, 91), ctx.synth_fill(n, 92)]
    lo, m = first + 100, 1200
    ref = ctx.to_host(whole.commit_batch_device(cols, n=m, first=lo))
    for p_ in (shard, loaded):
        got = ctx.to_host(p_.commit_batch_device(cols, n=m, first=lo))
        assert all((ffi.g1_to_affine(a) == ffi.g1_to_affine(b)).all() for a, b in zip(got, ref))
    with pytest.raises(ffi.ZkhipError):
        shard.commit_batch_device(cols, n=m, first=first - 1)      # leaves the shard, and the context has no communicator
    with pytest.raises(ffi.ZkhipError):
        shard.read_bases(shard.g, 0, 4)
    whole.free(); shard.free(); loaded.free()


def test_published_eip196_add_and_mul_vectors_hip(zk, oracle):
    """the HIP MSM (window tables built from the caller's bases, signed digits, bucket accumulation, tail) on go-ethereum's published
    bn256Add / bn256ScalarMul vectors "chfast1" — points of unknown discrete logarithm, results fixed by a third party
    (tests/golden/eip196_published.json): a + b and [s] P come out as published"""
    ffi, ctx = zk
    zo = oracle
    v = load("eip196_published.json")
    pt = lambda xy: (H(xy[0]), H(xy[1]))
    a, b, c = pt(v["add"]["a"]), pt(v["add"]["b"]), pt(v["add"]["sum"])
    res = aff(zk, ffi.best_multiexp(ctx, zo.fr_arr_from_ints([1, 1]), zo.affine_from_ints([a, b])))
    assert zo.affine_to_ints(res)[0] == c
    m, s, r = pt(v["mul"]["point"]), H(v["mul"]["scalar"]), pt(v["mul"]["product"])
    res = aff(zk, ffi.best_multiexp(ctx, zo.fr_arr_from_ints([s, 0, 5, 0]), zo.affine_from_ints([m, a, (0, 0), b])))      # zero scalars and an identity base contribute nothing
    assert zo.affine_to_ints(res)[0] == r
    R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
    for case in v["mul_more"]:
        pnt, sc, want = pt(case["point"]), H(case["scalar"]) % R, pt(case["product"])
        res = aff(zk, ffi.best_multiexp(ctx, zo.fr_arr_from_ints([sc, 1, R - 1]), zo.affine_from_ints([pnt, a, a])))
        assert zo.affine_to_ints(res)[0] == want
    res = aff(zk, ffi.best_multiexp(ctx, zo.fr_arr_from_ints([1, 1]), zo.affine_from_ints([pt(v["add2"]["a"]), pt(v["add2"]["b"])])))
    assert zo.affine_to_ints(res)[0] == pt(v["add2"]["sum"])
