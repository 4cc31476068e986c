"""GPU parity: the quotient on quotient_poly_degree cosets of the size-n domain (csrc/cosets.hip, the layout zkhip_create_proof_ex
works in when it saves rows) against the extended-domain entry points, which are themselves checked against the oracle and the golden
vectors in test_gpu_ntt.py / test_gpu_sweep.py.  Block r of a coset-layout column is the extended column at rows r + E i."""
import numpy as np
import pytest

import halo2_zkcert_amd.prover as pv
from test_gpu_sweep import _to_device_kw, random_circuit

pytestmark = pytest.mark.gpu


def _blocks(col, n, q):
    """extended-domain column (E n, 4) -> its first q cosets, block after block (q n, 4)"""
    e = col.shape[0] // n
    return np.ascontiguousarray(col.reshape(n, e, 4).transpose(1, 0, 2)[:q].reshape(q * n, 4))


@pytest.mark.parametrize("k,degree", [(4, 4), (6, 6), (9, 4), (12, 4), (13, 7), (16, 4)])
def test_coeff_to_cosets_is_the_extended_subset(zk, oracle, k, degree):
    """q size-n transforms of s_r^t a_t == the rows r + E i of coeff_to_extended (1-pass, 2-pass and register-tiled plans)"""
    ffi, ctx = zk
    dom = ffi.EvaluationDomain(ctx, degree, k)
    q, shifts = dom.cosets()
    assert q == degree - 1 and shifts.shape == (q, 4)
    polys = [ctx.synth_fill(1 << k, 7100 + 10 * k + i) for i in range(5)]
    ext = dom.coeff_to_extended_device(polys)
    cos = dom.coeff_to_cosets_device(polys)
    for e_, c_ in zip(ext, cos):
        assert (ctx.to_host(c_) == _blocks(ctx.to_host(e_), 1 << k, q)).all()
    # the generators: s_0 = g_coset, s_r = s_0 w_ext^r
    s = [pv.from_mont_host(r_) for r_ in shifts]
    assert s[0] == pv.from_mont_host(dom.g_coset)
    for r in range(1, q):
        assert s[r] == s[r - 1] * pv.from_mont_host(dom.extended_omega) % pv.R
    dom.free()


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
@pytest.mark.parametrize("k,degree", [(13, 4), (14, 6)])
def test_coset_transforms_on_every_tile_kernel(zk, oracle, k, degree, mode):
    """the per-index pre- / post-scaling tables of the coset transforms (NttScale::pre_tab / post_tab) through each family of tile kernels:
    stage-per-barrier (ntt_r8 = 0), 8 elements per thread (1), 4 per thread on 2048- (2) and 1024-element tiles (3) — forwards against the
    extended-domain transform (itself held against the oracle in test_gpu_ntt.py), backwards by recovering random pieces exactly"""
    ffi, ctx = zk
    n = 1 << k
    dom = ffi.EvaluationDomain(ctx, degree, k)
    q, shifts = dom.cosets()
    polys = [ctx.synth_fill(n, 7500 + 10 * k + i) for i in range(3)]
    ctx.set_option("ntt_r8", 4)
    ext = [ctx.to_host(e_) for e_ in dom.coeff_to_extended_device(polys)]
    ctx.set_option("ntt_r8", mode)
    try:
        cos = dom.coeff_to_cosets_device(polys)
        for e_, c_ in zip(ext, cos):
            assert (ctx.to_host(c_) == _blocks(e_, n, q)).all()
        pieces = [ctx.synth_fill(n, 7600 + 10 * k + j) for j in range(q)]
        pc = dom.coeff_to_cosets_device(pieces)
        blocks = []
        for r in range(q):
            c = pow(pv.from_mont_host(shifts[r]), n, pv.R)
            coeffs = np.stack([pv.fr_from_int_host((c - 1) * pow(c, j, pv.R)) for j in range(q)])
            blocks.append(ffi.linear_combination_device(ctx, [p_[r * n:(r + 1) * n] for p_ in pc], coeffs))
        import torch

        got = ctx.to_host(dom.cosets_to_pieces_device(torch.cat(blocks, dim=0).contiguous()))
        assert (got == np.concatenate([ctx.to_host(p_) for p_ in pieces])).all()
    finally:
        ctx.set_option("ntt_r8", 4)
        dom.free()


def test_cosets_refused_when_they_save_nothing(zk):
    ffi, ctx = zk
    for degree in (3, 5, 9):     # q = 2, 4, 8 = the extension factor
        dom = ffi.EvaluationDomain(ctx, degree, 6)
        with pytest.raises(ffi.ZkhipError):
            dom.cosets()
        dom.free()


@pytest.mark.parametrize("k,degree", [(4, 4), (5, 6), (10, 4), (12, 6), (13, 8), (15, 4)])
def test_cosets_to_pieces_recovers_the_quotient(zk, oracle, k, degree):
    """h with q n random coefficients -> numerator h (X^n - 1) on the q cosets (piece by piece: h(s w^i) = sum_j (s^n)^j h_j(s w^i))
    -> zkhip_cosets_to_pieces_device gives back exactly h: the per-coset inverse transforms and the q x q combination"""
    ffi, ctx = zk
    n = 1 << k
    dom = ffi.EvaluationDomain(ctx, degree, k)
    q, shifts = dom.cosets()
    pieces = [ctx.synth_fill(n, 7300 + 10 * k + j) for j in range(q)]
    pc = dom.coeff_to_cosets_device(pieces)
    blocks = []
    for r in range(q):
        c = pow(pv.from_mont_host(shifts[r]), n, pv.R)
        coeffs = np.stack([pv.fr_from_int_host((c - 1) * pow(c, j, pv.R)) for j in range(q)])
        blocks.append(ffi.linear_combination_device(ctx, [p_[r * n:(r + 1) * n] for p_ in pc], coeffs))
    import torch

    vals = torch.cat(blocks, dim=0).contiguous()
    got = ctx.to_host(dom.cosets_to_pieces_device(vals))
    want = np.concatenate([ctx.to_host(p_) for p_ in pieces])
    assert (got == want).all()
    dom.free()


@pytest.mark.parametrize("cfg", [
    dict(k=8, degree=4, bf=6, n_adv=4, n_fix=3, n_lookups=1, n_perm=6, seed=21),    # RSA-shaped: 3 of 4 cosets
    dict(k=4, degree=4, bf=3, n_adv=2, n_fix=2, n_lookups=1, n_perm=3, seed=22),    # 48 rows: one partial wave
    dict(k=5, degree=4, bf=3, n_adv=2, n_fix=2, n_lookups=1, n_perm=3, seed=23),    # 96 rows: half-wave blocks
    dict(k=7, degree=6, bf=5, n_adv=3, n_fix=2, n_lookups=2, n_perm=5, seed=24),    # 5 of 8 cosets, chunks of 4 columns
    dict(k=6, degree=8, bf=4, n_adv=3, n_fix=1, n_lookups=0, n_perm=4, seed=25),    # 7 of 8
])
def test_sweep_over_cosets_equals_extended_rows(zk, oracle, cfg):
    """zkhip_evaluate_h_cosets_device on the coset layout of the SAME columns == the rows r + E i of zkhip_evaluate_h_device (which
    test_gpu_sweep.py holds against the oracle): rotations wrap inside a block, the permutation argument's X value is s_r w^i.
    Row ranges of the q n rows glue together (the row-sharded multi-GPU sweep)."""
    ffi, ctx = zk
    zo = oracle
    dom_o, kw = random_circuit(zo, **cfg)
    n, q = 1 << cfg["k"], cfg["degree"] - 1
    pack = ffi.EvalhPack()
    pack.build(**_to_device_kw(ctx, kw))
    ext = ctx.to_host(ffi.evaluate_h(ctx, pack, dom_o.extended_n))
    ckw = dict(kw)
    for key in ("fixed", "advice", "instance", "sigma", "perm_z", "lookup_z", "lookup_a", "lookup_s"):
        ckw[key] = [_blocks(np.asarray(c), n, q) for c in kw[key]]
    for key in ("l0", "l_last", "l_active"):
        ckw[key] = _blocks(np.asarray(kw[key]), n, q)
    dom = ffi.EvaluationDomain(ctx, cfg["degree"], cfg["k"])
    cpack = ffi.EvalhPack()
    cpack.build(**_to_device_kw(ctx, ckw))
    got = ctx.to_host(dom.evaluate_h_cosets(cpack))
    assert (got == _blocks(ext, n, q)).all()
    if q * n >= 256:
        cuts = [0, 64, q * n // 2, q * n - 128, q * n]
        parts = [ctx.to_host(dom.evaluate_h_cosets(cpack, a, b - a)) for a, b in zip(cuts, cuts[1:])]
        assert (np.concatenate(parts) == got).all()
    with pytest.raises(ffi.ZkhipError):
        dom.evaluate_h_cosets(cpack, 16, q * n)
    dom.free()


@pytest.mark.parametrize("k", [6, 11])
def test_create_proof_same_bytes_on_either_domain(zk, oracle, k):
    """zkhip_create_proof_ex with the quotient on 3 cosets (the default for cs.degree() = 4) and on the extended domain
    (zkhip_set_option coset_quotient = 0): the same proof bytes, equal to the oracle backend's, and the verifier accepts them.  For a
    witness that does NOT satisfy the circuit the numerator is no multiple of X^n - 1: neither output is a proof (the verifier rejects
    both), the extended-domain bytes are upstream's, the coset ones differ from the quotient commitment on — documented in
    include/zkhip.h and DESIGN.md."""
    from oracle_backend import OracleBackend
    from verify_util import verify_proof

    ffi, ctx = zk
    sh = pv.CircuitShape.small(k)
    gp = pv.Prover(pv.GpuBackend(ctx, ffi), sh, satisfiable=True)
    cp = pv.Prover(OracleBackend(8), sh, satisfiable=True)
    w = gp.witness(3)
    on = gp.prove_native(w, transcript="poseidon")
    # on the coset path the key's extended cosets are not read: a caller may leave them out ...
    pk = gp._native_key()
    saved = {f: getattr(pk, f) for f in ("fixed_cosets", "sigma_cosets", "l0", "l_last", "l_active_row")}
    for f in saved:
        setattr(pk, f, None)
    try:
        assert gp.prove_native(w, transcript="poseidon")["proof"] == on["proof"]
        ctx.set_option("coset_quotient", 0)      # ... but the extended-domain path needs them and says so
        with pytest.raises(ffi.ZkhipError, match="extended cosets are missing"):
            gp.prove_native(w, transcript="poseidon", auto_extended=False)
    finally:
        ctx.set_option("coset_quotient", 1)
        for f, v in saved.items():
            setattr(pk, f, v)
    ctx.set_option("coset_quotient", 0)
    try:
        off = gp.prove_native(w, transcript="poseidon")
        assert on["proof"] == off["proof"] == cp.prove(cp.witness(3), transcript="poseidon")["proof"]
        assert verify_proof(gp, w, on["proof"], "poseidon")
        # an unsatisfied witness: a run of advice cells off
        bad = gp.witness(3)
        col = ctx.to_host(bad["advice"][0]).copy()
        col[1:40, 0] ^= np.uint64(1)
        bad["advice"][0] = ctx.to_device(col)
        off_bad = gp.prove_native(bad, transcript="poseidon")
        ctx.set_option("coset_quotient", 1)
        on_bad = gp.prove_native(bad, transcript="poseidon")
        assert not verify_proof(gp, bad, off_bad["proof"], "poseidon") and not verify_proof(gp, bad, on_bad["proof"], "poseidon")
        nq = [i for i, (tag, _) in enumerate(on_bad["commitments"]) if tag == "quotient"][0]
        assert on_bad["commitments"][:nq] == off_bad["commitments"][:nq]
    finally:
        ctx.set_option("coset_quotient", 1)


def test_key_release_frees_the_cached_columns(zk, oracle):
    """zkhip_key_release: the per-key caches (coset-layout fixed / sigma / l columns, sorted lookup table) go back to the allocator
    and a later proof with the same key rebuilds them — same bytes"""
    import torch

    ffi, ctx = zk
    k = 14
    gp = pv.Prover(pv.GpuBackend(ctx, ffi), pv.CircuitShape.small(k), satisfiable=True)
    w = gp.witness(1)
    first = gp.prove_native(w, transcript="poseidon")["proof"]
    torch.cuda.synchronize()
    held = torch.cuda.mem_get_info()[0]
    gp.release()
    freed = torch.cuda.mem_get_info()[0] - held
    sh = gp.shape
    cached = (len(gp.fixed_lagrange) + len(sh.perm_columns) + 3) * 3 * (32 << k)      # the key's columns on 3 cosets
    assert freed >= cached, (freed, cached)
    assert gp.prove_native(w, transcript="poseidon")["proof"] == first
    gp.release()


@pytest.mark.parametrize("k", [17, 20])
def test_full_size_proofs_agree_on_either_domain(zk, oracle, k):
    """BASELINE sizes: the RSA-shaped k = 17 circuit (two-pass size-n transforms) and the aggregation shape at k = 20 (2^20: two 10-bit
    passes; 3 x 2^20 sweep rows) through zkhip_create_proof_ex on 3 cosets and on the extended domain: identical proof bytes"""
    ffi, ctx = zk
    sh = pv.CircuitShape.rsa(k)
    gp = pv.Prover(pv.GpuBackend(ctx, ffi), sh, satisfiable=True)
    w = gp.witness(2)
    on = gp.prove_native(w, transcript="poseidon")["proof"]
    ctx.set_option("coset_quotient", 0)
    try:
        off = gp.prove_native(w, transcript="poseidon")["proof"]
    finally:
        ctx.set_option("coset_quotient", 1)
    assert on == off
    gp.release()
    del gp, w
