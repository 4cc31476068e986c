"""GPU parity: the whole create_proof-shaped pass (16 commitments + quotient pieces) vs the oracle backend."""
import hashlib
import json
import os

import numpy as np
import pytest

import halo2_zkcert_amd.prover as pv
from oracle_backend import OracleBackend

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k", [7, 10])
def test_pass_matches_oracle(zk, oracle, k):
    ffi, ctx = zk
    sh = pv.CircuitShape.small(k)
    gp = pv.Prover(pv.GpuBackend(ctx, ffi), sh)
    cp = pv.Prover(OracleBackend(8), sh)
    # keygen-side artefacts first
    for a, b in zip(gp.fixed_cosets + gp.sigma_cosets + [gp.l0, gp.l_last, gp.l_active],
                    cp.fixed_cosets + cp.sigma_cosets + [cp.l0, cp.l_last, cp.l_active]):
        assert (ctx.to_host(a) == b).all()
    tg = gp.prove(gp.witness(3))
    tc = cp.prove(cp.witness(3))
    assert tg["commitments"] == tc["commitments"]          # every commitment, byte for byte
    assert tg["challenges"] == tc["challenges"]
    assert len(tg["evals"]) == len(tc["evals"]) and all(ra == rb and (ea == eb).all() for (ra, ea), (rb, eb) in zip(tg["evals"], tc["evals"]))
    for a, b in zip(tg["h_pieces"], tc["h_pieces"]):
        assert (ctx.to_host(a) == b).all()


def test_sha_shape_pass_matches_oracle(zk, oracle):
    """BASELINE configs[2] shape (wide boolean-gate circuit, degree 5 -> extension 4, no lookup) at a size the oracle
    finishes in seconds: whole pass, byte for byte."""
    ffi, ctx = zk
    sh = pv.CircuitShape.sha256(9, n_advice=12, n_fixed=5)
    gp = pv.Prover(pv.GpuBackend(ctx, ffi), sh)
    cp = pv.Prover(OracleBackend(8), sh)
    tg = gp.prove(gp.witness(1))
    tc = cp.prove(cp.witness(1))
    assert tg["commitments"] == tc["commitments"] and tg["challenges"] == tc["challenges"]
    for a, b in zip(tg["h_pieces"], tc["h_pieces"]):
        assert (ctx.to_host(a) == b).all()
    assert len(tg["h_pieces"]) == 4 and gp.dom.extended_k == 11


def test_rsa_k17_pass_properties(zk, oracle):
    """BASELINE size (configs[1], RSA k=17): size-independent checks on the full pass.
    Determinism, and the quotient pieces commit to the same points the oracle derives from the
    pieces' own coefficients via the SRS trapdoor: commit(piece) == [piece(s)] G."""
    ffi, ctx = zk
    zo = oracle
    sh = pv.CircuitShape.rsa(17)
    s = 0x1D5C0FFEE
    gp = pv.Prover(pv.GpuBackend(ctx, ffi), sh, srs_trapdoor=s)
    w = gp.witness(0)
    t1 = gp.prove(w)
    t2 = gp.prove(w)
    assert t1["commitments"] == t2["commitments"] and t1["n_commitments"] == 16
    qc = [c for tag, c in t1["commitments"] if tag == "quotient"]
    sm = zo.fr_from_int(s)
    for piece, c in zip(t1["h_pieces"], qc):
        exp = zo.g1_mul_gen(zo.eval_polynomial(ctx.to_host(piece), sm))
        assert zo.g1_to_bytes(exp).hex() == c
    # advice commitments (Lagrange basis) against the trapdoor too: commit_lagrange(v) == [iNTT(v)(s)] G
    dom = gp.dom
    col = w["advice"][0].clone()
    dom.lagrange_to_coeff_device([col])
    exp = zo.g1_mul_gen(zo.eval_polynomial(ctx.to_host(col), sm))
    assert zo.g1_to_bytes(exp).hex() == t1["commitments"][0][1]


@pytest.mark.parametrize("k", [6, 9])
def test_satisfiable_proof_verifies_and_matches_oracle(zk, oracle, k):
    """A satisfiable instance end to end on the GPU: the proof passes oracle/pyref.py's verifier equations (gates, permutation,
    lookup, quotient, SHPLONK under the SRS trapdoor), equals the oracle backend's proof byte for byte, and a corrupted
    evaluation is rejected."""
    from verify_util import verify_trace

    ffi, ctx = zk
    sh = pv.CircuitShape.small(k)
    gp = pv.Prover(pv.GpuBackend(ctx, ffi), sh, satisfiable=True)
    cp = pv.Prover(OracleBackend(8), sh, satisfiable=True)
    wg, wc = gp.witness(2), cp.witness(2)
    for a, b in zip(wg["advice"] + gp.sigma_lagrange + gp.fixed_lagrange, wc["advice"] + cp.sigma_lagrange + cp.fixed_lagrange):
        assert (ctx.to_host(a) == b).all()
    tg, tc = gp.prove(wg), cp.prove(wc)
    assert tg["commitments"] == tc["commitments"] and tg["challenges"] == tc["challenges"]
    assert verify_trace(gp, wg, tg)

    def bad_eval(evals, coms, instance):
        key = (("lookup_a", 0), -1)
        evals[key] = (evals[key] + 1) % pv.R
    assert not verify_trace(gp, wg, tg, tamper=bad_eval)


def test_rsa_k17_valid_proof(zk, oracle):
    """BASELINE size: the RSA-shaped k = 17 pass on a satisfiable instance produces a proof the verifier accepts."""
    from verify_util import verify_trace

    ffi, ctx = zk
    sh = pv.CircuitShape.rsa(17)
    gp = pv.Prover(pv.GpuBackend(ctx, ffi), sh, satisfiable=True)
    w = gp.witness(0)
    t = gp.prove(w)
    assert t["n_commitments"] == 16
    assert verify_trace(gp, w, t)


@pytest.mark.parametrize("k", [6, 10])
def test_native_create_proof_matches_schedule(zk, oracle, k):
    """zkhip_create_proof (the whole schedule in the library, transcript through callbacks) produces the same proof as the Python
    schedule over the small entry points — commitments, challenges, evaluations, quotient — and the verifier accepts it."""
    from verify_util import verify_trace

    ffi, ctx = zk
    sh = pv.CircuitShape.small(k)
    gp = pv.Prover(pv.GpuBackend(ctx, ffi), sh, satisfiable=True)
    w = gp.witness(4)
    ta = gp.prove(w)
    tb = gp.prove_native(w, fetch_h=True)
    tc = gp.prove_native(w, python_transcript=True)
    assert tb["commitments"] == tc["commitments"] and tb["challenges"] == tc["challenges"]
    # the proof bytes: every commitment (32 B compressed) and evaluation (32 B canonical) in transcript order
    nq = len(tb["evals"]) - 1
    assert len(tb["proof"]) == 32 * (tb["n_commitments"] + nq)
    assert ta["commitments"] == tb["commitments"]
    assert ta["challenges"] == tb["challenges"]
    assert [q for q, _ in ta["evals"]] == [q for q, _ in tb["evals"]]
    assert all((ea == eb).all() for (_, ea), (_, eb) in zip(ta["evals"], tb["evals"]))
    for a, b in zip(ta["h_pieces"], tb["h_pieces"]):
        assert (ctx.to_host(a) == b).all()
    assert verify_trace(gp, w, tb)


def test_native_create_proof_rsa_k17(zk, oracle):
    from verify_util import verify_trace

    ffi, ctx = zk
    gp = pv.Prover(pv.GpuBackend(ctx, ffi), pv.CircuitShape.rsa(17), satisfiable=True)
    w = gp.witness(0)
    t = gp.prove_native(w)
    assert t["n_commitments"] == 16
    assert t["commitments"] == gp.prove(w)["commitments"]
    assert verify_trace(gp, w, t)


def test_native_create_proof_sha_shape(zk, oracle):
    """no lookups, one permutation set, degree 5: the native schedule against the Python one"""
    ffi, ctx = zk
    sh = pv.CircuitShape.sha256(9, n_advice=12, n_fixed=5)
    gp = pv.Prover(pv.GpuBackend(ctx, ffi), sh)
    w = gp.witness(1)
    ta, tb = gp.prove(w), gp.prove_native(w)
    assert ta["commitments"] == tb["commitments"] and ta["challenges"] == tb["challenges"]
    assert all((ea == eb).all() for (_, ea), (_, eb) in zip(ta["evals"], tb["evals"]))


def test_native_create_proof_reports_unsatisfiable_lookup(zk, oracle):
    """a lookup input that is not in the table: ConstraintSystemFailure from both schedules, nothing half-written afterwards"""
    ffi, ctx = zk
    zo = oracle
    sh = pv.CircuitShape.small(7)
    gp = pv.Prover(pv.GpuBackend(ctx, ffi), sh, satisfiable=True)
    w = gp.witness(0)
    good = gp.prove_native(w)["commitments"]
    bad = dict(w)
    col = w["advice"][sh.n_basic].clone()
    col[3] = ctx.to_device(zo.fr_arr_from_ints([0x123456789]))[0]
    bad["advice"] = w["advice"][:sh.n_basic] + [col] + w["advice"][sh.n_basic + 1:]
    with pytest.raises(ffi.ConstraintSystemFailure):
        gp.prove_native(bad)
    with pytest.raises(ffi.ConstraintSystemFailure):
        gp.prove(bad)
    assert gp.prove_native(w)["commitments"] == good


def test_agg_k22_pass_properties(zk, oracle):
    """BASELINE configs[3] size (k = 22, the aggregation circuit's): one zkhip_create_proof pass on a satisfiable instance.
    Size-independent checks: the quotient pieces' commitments equal [piece(s)] G from the pieces' own coefficients (SRS trapdoor),
    the proof is deterministic, and its byte length is what the transcript order dictates."""
    ffi, ctx = zk
    zo = oracle
    s = 0x1D5C0FFEE
    gp = pv.Prover(pv.GpuBackend(ctx, ffi), pv.CircuitShape.rsa(22), srs_trapdoor=s, satisfiable=True)
    w = gp.witness(0)
    t1 = gp.prove_native(w, fetch_h=True)
    sm = zo.fr_from_int(s)
    qc = [c for tag, c in t1["commitments"] if tag == "quotient"]
    assert len(qc) == 3
    for piece, c in zip(t1["h_pieces"], qc):
        assert zo.g1_to_bytes(zo.g1_mul_gen(zo.eval_polynomial(piece, sm))).hex() == c
    t2 = gp.prove_native(w)
    assert t2["proof"] == t1["proof"] and len(t1["proof"]) == 32 * (t1["n_commitments"] + len(t1["evals"]) - 1)
    del gp, w


def test_native_create_proof_two_expression_lookup(zk, oracle):
    """A lookup with two input and two table expressions: the theta-compression passes run (no single-column shortcut, no cached
    table); the native schedule equals the Python one and the oracle backend's."""
    ffi, ctx = zk
    k = 8
    sh = pv.CircuitShape.small(k)
    A = lambda c, r: ("advice", c, r)
    sh.lookups = [([A(sh.n_basic, 0), A(0, 0)], [("fixed", sh.n_fixed - 1, 0), ("fixed", sh.n_basic, 0)])]
    sh._queries = None
    gp = pv.Prover(pv.GpuBackend(ctx, ffi), sh)
    cp = pv.Prover(OracleBackend(8), sh)
    n = 1 << k
    idx = (np.arange(n, dtype=np.int64) * 37 + 11) % (n // 2)

    def wit(p):   # (input_1, input_2)[row] = (table_1, table_2)[idx[row]]: a satisfiable two-column lookup
        w = p.witness(5)
        w["advice"][sh.n_basic] = p.b.gather(p.fixed_lagrange[sh.n_fixed - 1], idx)
        w["advice"][0] = p.b.gather(p.fixed_lagrange[sh.n_basic], idx)
        return w

    wg, wc = wit(gp), wit(cp)
    # the gates are not satisfied by this witness (only the lookup is): byte parity of the quotient needs the extended domain
    # (include/zkhip.h, the coset layout's note; test_gpu_cosets.py)
    ctx.set_option("coset_quotient", 0)
    try:
        ta, tb, tc = gp.prove(wg), gp.prove_native(wg), cp.prove(wc)
    finally:
        ctx.set_option("coset_quotient", 1)
    assert ta["commitments"] == tb["commitments"] == tc["commitments"]
    assert ta["challenges"] == tb["challenges"] == tc["challenges"]


def test_native_create_proof_evm_transcript(zk, oracle):
    """zkhip_create_proof over the library's EvmTranscript (Keccak-256): a different Fiat-Shamir, so different challenges, but the
    first-phase commitments are the same points, the proof has the EVM layout (64-byte points, 32-byte scalars), the quotient
    commitments satisfy the SRS-trapdoor identity, and the SHPLONK opening verifies for the Keccak challenges."""
    import pyref as P

    ffi, ctx = zk
    zo = oracle
    s = 0x1D5C0FFEE
    sh = pv.CircuitShape.small(7)
    gp = pv.Prover(pv.GpuBackend(ctx, ffi), sh, srs_trapdoor=s, satisfiable=True)
    w = gp.witness(1)
    tb = gp.prove_native(w)
    te = gp.prove_native(w, evm=True, fetch_h=True)
    assert te["challenges"]["theta"] != tb["challenges"]["theta"]
    nq = len(te["evals"]) - 1
    assert len(te["proof"]) == 64 * te["n_commitments"] + 32 * nq
    adv_b = [np.asarray(p_) for p_ in tb["points"]["advice"]]
    adv_e = [np.asarray(p_) for p_ in te["points"]["advice"]]
    assert all((a == b).all() for a, b in zip(adv_b, adv_e))          # advice commitments precede every challenge
    sm = zo.fr_from_int(s)
    for piece, pt in zip(te["h_pieces"], te["points"]["quotient"]):
        assert zo.g1_to_bytes(zo.g1_mul_gen(zo.eval_polynomial(piece, sm))) == zo.g1_to_bytes(np.asarray(pt, dtype=np.uint64))
    # the multi-open under the Keccak challenges: pyref's verifier equation with this trace's points / evaluations
    from verify_util import _pts
    coms = {}
    for i, p_ in enumerate(_pts(te["points"]["advice"])):
        coms[("advice", i)] = p_
    L = len(sh.lookups)
    lp = _pts(te["points"].get("lookup_permuted", []))
    for i in range(L):
        coms[("lookup_a", i)], coms[("lookup_s", i)] = lp[i], lp[L + i]
    prods = _pts(te["points"]["products"])
    for i in range(sh.n_perm_sets):
        coms[("perm_z", i)] = prods[i]
    for i in range(L):
        coms[("lookup_z", i)] = prods[sh.n_perm_sets + i]
    coms[("random", 0)] = _pts(te["points"]["random_poly"])[0]
    for i, c in enumerate(gp.b.commit(gp.fixed_coeff, lagrange=False)):
        coms[("fixed", i)] = _pts([c[0]])[0]
    for i, c in enumerate(gp.b.commit(gp.sigma_coeff, lagrange=False)):
        coms[("sigma", i)] = _pts([c[0]])[0]
    vk = dict(k=sh.k, degree=sh.degree, blinding_factors=sh.blinding_factors, gates=sh.gates, lookups=sh.lookups, perm_columns=sh.perm_columns)
    evals = {q: v for q, v in pv.eval_ints(te).items() if q[0] != ("h", 0)}
    instance = [zo.fr_arr_to_ints(gp.b.to_host(c)) for c in w["instance"]]
    h1, h2 = _pts(te["points"]["shplonk_h1"])[0], _pts(te["points"]["shplonk_h2"])[0]
    assert P.plonk_verify(vk, instance, coms, _pts(te["points"]["quotient"]), evals, te["query_list"], te["challenges"], h1, h2, s)


def _cli():
    import importlib.util
    import os

    spec = importlib.util.spec_from_file_location("zkcert_cli", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "zkcert_cli.py"))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    return cli


def test_cli_shaped_driver_and_srs_file_round_trip(zk, tmp_path, capsys):
    """tools/zkcert_cli.py (the reference CLI's command / argument names, /root/reference/src/bin/cli.rs:95-211): the first run
    generates the SRS and leaves kzg_bn254_<k>.synthetic.srs under --params-path and writes the proving key (`gen_pk(.., Some(path))`)
    and the witness; the second run READS the SRS, the proving key file (`read_pk`) and the witness back — same proof bytes, as a bincode
    Snark (`gen_snark_shplonk(.., Some(path))`); the aggregation command reads the leaf snarks (`read_snark`) and takes its advice column
    count from the break points file; the EVM command's proof has the 64-byte point layout."""
    import json
    import os

    import halo2_zkcert_amd.formats as fm

    cli = _cli()
    params = str(tmp_path / "params")
    pk, wit = str(tmp_path / "rsa_1.pk"), str(tmp_path / "rsa_1.witness.npz")
    outs = []
    for name in ("a.proof", "b.proof"):
        cli.main(["prove-rsa", "--k", "9", "--params-path", params, "--pk-path", pk, "--witness-path", wit, "--proof-path", str(tmp_path / name)])
        outs.append(json.loads(capsys.readouterr().out.strip().splitlines()[-1]))
    assert not os.path.exists(os.path.join(params, "kzg_bn254_9.srs"))      # the reference's file name is never written with a synthetic SRS
    srs = os.path.join(params, "kzg_bn254_9.synthetic.srs")
    assert os.path.getsize(srs) == 4 + 2 * 512 * 64 + 256 and outs[0]["params"] == srs and outs[0]["transcript"] == "poseidon"
    assert open(srs, "rb").read()[-256:] != bytes(256)                      # real g2 / [s] g2, not identity points
    assert outs[0]["proving_key"]["source"] == "generated" and outs[1]["proving_key"]["source"] == "file" and os.path.getsize(pk) > 512 * 32 * 10
    a, b = fm.SnarkFile.read(tmp_path / "a.proof", protocol_len=0), fm.SnarkFile.read(tmp_path / "b.proof", protocol_len=0)
    assert a.proof == b.proof and a.instances == b.instances and len(a.proof) == outs[0]["proof_bytes"] and len(a.proof) % 32 == 0
    assert [len(c) for c in a.instances] == [32]
    fm.write_break_points(tmp_path / "bp.json", [[500, 501, 502, 503], []])          # 4 break points = 5 advice columns
    cli.main(["gen-x509-agg-evm-proof", "--agg-k", "9", "--params-path", params, "--no-pk-file", "--agg-proof-path", str(tmp_path / "e.proof"),
              "--snark-paths", str(tmp_path / "a.proof"), str(tmp_path / "b.proof"), "--break-points-path", str(tmp_path / "bp.json")])
    e = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert e["transcript"] == "evm-keccak" and e["proof_bytes"] > len(a.proof) and e["break_points"]["advice_columns"] == 5
    assert [s_["proof_bytes"] for s_ in e["snarks"]] == [len(a.proof)] * 2 and "a5+1" in e["circuit"]
    assert (tmp_path / "e.proof").stat().st_size == e["proof_bytes"]


def test_proof_from_an_oracle_written_proving_key_file(zk, oracle, tmp_path, capsys):
    """the artefact readers have a consumer.  The ORACLE backend runs the keygen-shaped setup, writes the proving key
    (formats.ProvingKeyFile: ProvingKey::write layout) and the witness, and proves on the CPU; the GPU side builds NOTHING of the key —
    `read_pk` -> zk_proving_key (fixed / sigma columns in three forms, l-polynomials from the file) — and must produce the oracle's
    proof bytes: through Prover(key_file=...) on both quotient paths, and through the CLI driver."""
    import json

    import halo2_zkcert_amd.formats as fm

    ffi, ctx = zk
    zo = oracle
    sh = pv.CircuitShape.rsa(9)
    cp = pv.Prover(OracleBackend(8), sh, satisfiable=True)
    w = cp.witness(5)
    ref = cp.prove(w, transcript="poseidon")["proof"]
    fixed_c = [c[0] for c in cp.b.commit(cp.fixed_coeff, lagrange=False)]
    sigma_c = [c[0] for c in cp.b.commit(cp.sigma_coeff, lagrange=False)]
    pk_path, wit_path = tmp_path / "rsa_1.pk", tmp_path / "rsa_1.witness.npz"
    fm.ProvingKeyFile.from_prover(cp, fixed_c, sigma_c).write(pk_path)
    cp.save_witness(w, wit_path)
    kf = fm.ProvingKeyFile.read(pk_path, n_perm_columns=len(sh.perm_columns), n_selectors=0)
    gp = pv.Prover(pv.GpuBackend(ctx, ffi), sh, key_file=kf)
    assert gp.key_source == "file" and not hasattr(gp, "_value_src")          # no keygen ran on this side
    gw = gp.load_witness(wit_path)
    assert gp.prove_native(gw, transcript="poseidon")["proof"] == ref           # quotient on cosets (key columns derived from fixed_polys / permutation polys)
    ctx.set_option("coset_quotient", 0)
    try:
        assert gp.prove_native(gw, transcript="poseidon")["proof"] == ref       # extended domain: the file's fixed_cosets / permutation cosets / l0 ...
    finally:
        ctx.set_option("coset_quotient", 1)
    # the vk's commitments in the file (made by the oracle's MSM) are what the GPU commits to for the same polynomials
    g_fixed = [np.asarray(c[0], dtype=np.uint64) for c in gp.b.commit(gp.fixed_coeff, lagrange=False)]
    assert (np.stack(g_fixed) == kf.fixed_commitments).all()
    gp.release()
    cli = _cli()
    cli.main(["prove-rsa", "--k", "9", "--params-path", str(tmp_path / "params"), "--pk-path", str(pk_path), "--witness-path", str(wit_path),
              "--proof-path", str(tmp_path / "cli.proof")])
    out = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert out["proving_key"]["source"] == "file"
    assert fm.SnarkFile.read(tmp_path / "cli.proof", protocol_len=0).proof == ref


# ------------------------------------------------------------------ round 2
@pytest.mark.parametrize("kind", ["poseidon", "blake2b", "evm"])
@pytest.mark.parametrize("k", [6, 9])
def test_native_proof_bytes_verify(zk, oracle, k, kind):
    """zkhip_create_proof_ex under each of the library's transcripts: the proof BYTES pass the byte-driven verifier (upstream's read
    order, challenges re-derived by an independent Python transcript), equal the Python schedule's bytes driven through the same
    library transcript, and — for the oracle backend under Poseidon — the CPU oracle's bytes."""
    from verify_util import verify_proof

    ffi, ctx = zk
    sh = pv.CircuitShape.small(k)
    gp = pv.Prover(pv.GpuBackend(ctx, ffi), sh, satisfiable=True)
    w = gp.witness(3)
    tn = gp.prove_native(w, transcript=kind)
    assert verify_proof(gp, w, tn["proof"], kind)
    tp = gp.prove(w, transcript=kind)
    assert tp["proof"] == tn["proof"]
    if kind == "poseidon":
        cp = pv.Prover(OracleBackend(8), sh, satisfiable=True)
        assert cp.prove(cp.witness(3), transcript=kind)["proof"] == tn["proof"]
    bad = bytearray(tn["proof"])
    bad[-70] ^= 4
    assert not verify_proof(gp, w, bytes(bad), kind)


def test_rsa_k17_poseidon_proof_bytes_verify(zk, oracle):
    """BASELINE configs[1] under the transcript the reference's prove-rsa really uses (gen_snark_shplonk = Poseidon,
    /root/reference/src/bin/cli.rs:320): the k = 17 proof's bytes verify — against a verifying key whose fixed / sigma commitments were
    made by the CPU ORACLE (its own SRS from the trapdoor, its own best_multiexp), so nothing the verifier is given comes from the GPU
    except the proof; the GPU-side commitments of the same polynomials are the same points."""
    from verify_util import verify_proof, vk_commitments

    ffi, ctx = zk
    gp = pv.Prover(pv.GpuBackend(ctx, ffi), pv.CircuitShape.rsa(17), satisfiable=True)
    w = gp.witness(0)
    t = gp.prove_native(w, transcript="poseidon")
    assert t["n_commitments"] == 16 and verify_proof(gp, w, t["proof"], "poseidon", oracle_vk=True)
    assert vk_commitments(gp, oracle_side=True) == vk_commitments(gp)
    bad = bytearray(t["proof"])
    bad[-70] ^= 4
    assert not verify_proof(gp, w, bytes(bad), "poseidon", oracle_vk=True)


def _cpu_oracle_digest(key):
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cpu_oracle_proof_digests.json")) as f:
        return json.load(f)["digests"][key]


def test_rsa_k17_proof_bytes_equal_the_cpu_oracle(zk, oracle, cpu_rsa17_proof):
    """north_star's own sentence at a BASELINE size: "proof bytes bit-identical to the reference CPU prover on the same SRS and witness".
    BASELINE configs[1] (RSA k = 17, the c = 16 window, the 2^17 NTT pass plan — kernels no k <= 12 case selects) under the transcript
    prove-rsa uses (/root/reference/src/bin/cli.rs:320, helpers.rs:233): the one-call GPU proof == the CPU oracle backend's proof, byte for
    byte, on the same key, witness and blinding draws; also with the advice columns handed over as host arrays."""
    ffi, ctx = zk
    sh = pv.CircuitShape.rsa(17)
    gp = pv.Prover(pv.GpuBackend(ctx, ffi), sh, satisfiable=True)
    wg = gp.witness(0)
    want = cpu_rsa17_proof      # Prover(OracleBackend, rsa(17)).prove(witness(0), "poseidon"): conftest.py, once per session
    got = bytes(gp.prove_native(wg, transcript="poseidon")["proof"])
    assert len(want) > 1000 and got == want
    assert hashlib.sha256(want).hexdigest() == _cpu_oracle_digest("rsa_k17/poseidon/witness0")      # the recorded digest is this oracle's
    assert bytes(gp.prove_native(wg, transcript="poseidon", host_inputs=True)["proof"]) == want
    gp.release()
    gp.b.params.free()


def test_sha_satisfiable_matches_oracle_k11(zk, oracle):
    """BASELINE configs[2] shape (32 advice / 12 fixed columns, degree 5, no lookup) as a SATISFIABLE instance at a size the oracle
    finishes in seconds: the GPU proof equals the oracle backend's byte for byte and verifies; a flipped bit in a bit column fails."""
    from verify_util import verify_proof

    ffi, ctx = zk
    zo = oracle
    sh = pv.CircuitShape.sha256(11)
    gp = pv.Prover(pv.GpuBackend(ctx, ffi), sh, satisfiable=True)
    cp = pv.Prover(OracleBackend(8), sh, satisfiable=True)
    wg, wc = gp.witness(1), cp.witness(1)
    for a, b in zip(wg["advice"], wc["advice"]):
        assert (ctx.to_host(a) == b).all()
    tg = gp.prove_native(wg, transcript="poseidon")
    tc = cp.prove(wc, transcript="poseidon")
    assert tg["proof"] == tc["proof"]
    assert verify_proof(gp, wg, tg["proof"], "poseidon")
    bad = dict(wg)
    col = wg["advice"][0].clone()
    col[6] = ctx.to_device(zo.fr_arr_from_ints([2]))[0]          # row 6 is active: b (1 - b) != 0
    bad["advice"] = [col] + wg["advice"][1:]
    assert not verify_proof(gp, bad, gp.prove_native(bad, transcript="poseidon")["proof"], "poseidon")


def test_sha_k19_satisfiable_proof_verifies(zk, oracle):
    """BASELINE configs[2] at FULL size (zkevm SHA256 shape, k = 19: 40 MSM_2^19 under the c = 17 window, 34 + 34 NTTs, the
    2^21-row sweep of 32 degree-<=5 gates): one zkhip_create_proof_ex pass on a satisfiable instance under the reference's
    transcript (Poseidon, /root/reference/src/bin/cli.rs:369); the proof bytes pass the byte-driven verifier, the quotient commitments
    satisfy the SRS-trapdoor identity, and the proof is deterministic."""
    from verify_util import verify_proof

    ffi, ctx = zk
    zo = oracle
    s = 0x1D5C0FFEE
    gp = pv.Prover(pv.GpuBackend(ctx, ffi), pv.CircuitShape.sha256(19), srs_trapdoor=s, satisfiable=True)
    assert gp.b.params.window() == (17, 15)
    w = gp.witness(0)
    t1 = gp.prove_native(w, transcript="poseidon", fetch_h=True)
    assert t1["n_commitments"] == 40
    assert verify_proof(gp, w, t1["proof"], "poseidon")
    assert hashlib.sha256(bytes(t1["proof"])).hexdigest() == _cpu_oracle_digest("sha256_k19/poseidon/witness0")      # == the CPU oracle's bytes (recorded digest)
    sm = zo.fr_from_int(s)
    qc = [c for tag, c in t1["commitments"] if tag == "quotient"]
    assert len(qc) == 4
    for piece, c in zip(t1["h_pieces"], qc):
        assert zo.g1_to_bytes(zo.g1_mul_gen(zo.eval_polynomial(piece, sm))).hex() == c
    assert gp.prove_native(w, transcript="poseidon")["proof"] == t1["proof"]
    # the 32 advice columns as pinned HOST arrays: uploaded in 4 groups on a copy stream, each group committed while the next is on the wire
    assert gp.prove_native(w, transcript="poseidon", host_inputs=True)["proof"] == t1["proof"]
    # ... and as PAGEABLE host arrays — what a Rust caller's Vec<Fr> columns are (/root/reference/src/helpers.rs:233: gen_snark_shplonk hands create_proof the witness as
    # vectors): their copies are issued from a worker thread (option host_copy_thread; a pageable hipMemcpyAsync blocks its caller), optionally from registered memory
    # (host_register); the same bytes every way, twice in a row, and with FRESH host arrays per proof
    for thr, reg in ((1, 0), (1, 0), (0, 0), (0, 1), (1, 1)):
        ctx.set_option("host_copy_thread", thr)
        ctx.set_option("host_register", reg)
        w.pop("advice_host_pageable", None)
        assert gp.prove_native(w, transcript="poseidon", host_inputs="pageable")["proof"] == t1["proof"], (thr, reg)
    ctx.set_option("host_copy_thread", 1)
    ctx.set_option("host_register", 0)
    gp.b.params.free()
    del gp, w


def test_caller_supplied_blinding_and_host_inputs(zk, oracle):
    """zkhip_create_proof_ex with the caller's rng draws (upstream's create_proof takes them from its `rng` argument) and with host
    advice columns (the Vec<Fr> a Rust caller holds):
      * blinding buffers equal to what the seeded generator produces -> the same proof bytes, from host and from device buffers;
      * other blinding values -> a different proof that still verifies; host advice / library-built instance columns -> same bytes."""
    from verify_util import verify_proof

    ffi, ctx = zk
    sh = pv.CircuitShape.small(8)
    gp = pv.Prover(pv.GpuBackend(ctx, ffi), sh, satisfiable=True)
    w = gp.witness(2)
    base, bf, L, Zp, n = w["base"], sh.blinding_factors, len(sh.lookups), sh.n_perm_sets, 1 << sh.k
    ref = gp.prove_native(w, transcript="poseidon")["proof"]
    lp = []
    for i in range(L):
        lp += [ctx.to_host(ctx.synth_fill(bf + 1, base + 300 + i)), ctx.to_host(ctx.synth_fill(bf + 1, base + 320 + i))]
    host = dict(lookup_permuted=np.concatenate(lp), perm_z=ctx.to_host(ctx.synth_fill(Zp * bf, base + 340)),
                lookup_z=ctx.to_host(ctx.synth_fill(L * bf, base + 360)), random_poly=ctx.to_host(ctx.synth_fill(n, base + 380)))
    assert gp.prove_native(w, transcript="poseidon", blinding=host)["proof"] == ref
    dev = {k_: ctx.to_device(v) for k_, v in host.items()}
    assert gp.prove_native(w, transcript="poseidon", blinding=dev)["proof"] == ref
    assert gp.prove_native(w, transcript="poseidon", host_inputs=True)["proof"] == ref
    assert gp.prove_native(w, transcript="poseidon", host_inputs=True, blinding=host)["proof"] == ref
    other = dict(host)
    other["perm_z"] = ctx.to_host(ctx.synth_fill(Zp * bf, 987654321))
    other["random_poly"] = ctx.to_host(ctx.synth_fill(n, 123456789))
    t = gp.prove_native(w, transcript="poseidon", blinding=other)
    assert t["proof"] != ref and verify_proof(gp, w, t["proof"], "poseidon")


def test_options_are_context_state(zk):
    """tuning knobs: read from the environment once (zkhip_init), changed with zkhip_set_option, unknown names rejected"""
    ffi, ctx = zk
    ctx.set_option("msm_debug", 0)
    ctx.set_option("ZKHIP_LATE_OVERLAP", -1)
    with pytest.raises(ffi.ZkhipError):
        ctx.set_option("no_such_knob", 1)
    ctx.trim()


def test_two_lookups_native_matches_oracle(zk, oracle):
    """two lookup arguments (interleaved permuted commitments, two lookup grand products, 10 lookup evaluations): zkhip_create_proof_ex
    equals the oracle backend byte for byte and the bytes verify"""
    from verify_util import verify_proof

    ffi, ctx = zk
    sh = pv.CircuitShape("two_lookups_k7", 7, 2, 2, 1, 4, 6, 0x2100C7)
    gp = pv.Prover(pv.GpuBackend(ctx, ffi), sh, satisfiable=True)
    cp = pv.Prover(OracleBackend(8), sh, satisfiable=True)
    w = gp.witness(0)
    t = gp.prove_native(w, transcript="poseidon")
    assert t["proof"] == cp.prove(cp.witness(0), transcript="poseidon")["proof"]
    assert verify_proof(gp, w, t["proof"], "poseidon")


@pytest.mark.parametrize("degree", [3, 6, 9])
def test_degree_and_extension_factors(zk, oracle, degree):
    """cs.degree() 3 / 6 / 9 -> extension factors 2 / 8 / 8 with 2 / 5 / 8 quotient pieces and permutation chunks of 1 / 4 / 7 columns:
    a product gate of that degree over rotations, no lookup; native schedule == Python schedule == oracle backend (bytes)."""
    ffi, ctx = zk
    A = lambda c, r: ("advice", c, r)
    g = ("fixed", 0, 0)
    for i in range(degree - 1):
        g = ("prod", g, A(i % 3, (i % 4) - 1))
    sh = pv.CircuitShape(f"deg{degree}_k6", 6, 3, 0, 1, degree, 6, 0xDE600 + degree, gates=[g, ("sum", A(0, 0), ("neg", A(1, 1)))], n_fixed=2,
                        perm_columns=[("advice", 0), ("advice", 1), ("advice", 2), ("instance", 0)])
    gp = pv.Prover(pv.GpuBackend(ctx, ffi), sh)
    cp = pv.Prover(OracleBackend(8), sh)
    assert gp.dom.quotient_poly_degree == degree - 1 and (1 << (gp.dom.extended_k - 6)) >= degree - 1
    w = gp.witness(2)
    # a random witness does not satisfy the gates: the extended domain for byte parity (degree 6 would otherwise run on 5 of 8 cosets;
    # that path is checked piece by piece in test_gpu_cosets.py)
    ctx.set_option("coset_quotient", 0)
    try:
        ta, tb, tc = gp.prove(w, transcript="blake2b"), gp.prove_native(w), cp.prove(cp.witness(2), transcript="blake2b")
    finally:
        ctx.set_option("coset_quotient", 1)
    assert ta["proof"] == tb["proof"] == tc["proof"]
    assert len([1 for tag, _ in tb["commitments"] if tag == "quotient"]) == degree - 1


@pytest.mark.parametrize("k", [4, 5])
def test_smallest_circuits(zk, oracle, k):
    """n = 16 and 32 rows (9 and 25 usable): every kernel at its smallest size — single-pass NTTs, MSMs of 16 points (window 3),
    one-tile sorts — native == oracle, and the proof verifies"""
    from verify_util import verify_proof

    ffi, ctx = zk
    sh = pv.CircuitShape.small(k)
    gp = pv.Prover(pv.GpuBackend(ctx, ffi), sh, satisfiable=True)
    cp = pv.Prover(OracleBackend(2), sh, satisfiable=True)
    w = gp.witness(0)
    t = gp.prove_native(w, transcript="poseidon")
    assert t["proof"] == cp.prove(cp.witness(0), transcript="poseidon")["proof"]
    assert verify_proof(gp, w, t["proof"], "poseidon")


def test_agg_k22_evm_proof_bytes_verify(zk, oracle):
    """BASELINE configs[3] at FULL size, exactly the headline configuration of bench.py: the aggregation-shaped k = 22 circuit (3 + 1 advice
    columns, range lookup with lookup_bits = 21, /root/reference/src/bin/cli.rs:475) proved by one zkhip_create_proof_ex call under the
    Keccak EvmTranscript (gen_evm_proof_shplonk, cli.rs:519); the proof BYTES (64-byte big-endian points, 32-byte scalars) pass the
    byte-driven verifier, and the caller's-rng path (host blinding buffers, host advice columns) gives another valid proof."""
    from verify_util import verify_proof

    ffi, ctx = zk
    sh = pv.CircuitShape.agg(22)
    gp = pv.Prover(pv.GpuBackend(ctx, ffi), sh, satisfiable=True)
    assert gp.b.params.window() == (20, 13)
    w = gp.witness(0)
    t = gp.prove_native(w, transcript="evm")
    assert t["n_commitments"] == 16 and len(t["proof"]) == 64 * 16 + 32 * (len(t["evals"]) - 1)
    assert verify_proof(gp, w, t["proof"], "evm")
    # ... and they are the bytes the CPU oracle produced for this instance (north_star: bit-identical to the CPU prover on the same SRS and
    # witness): the recorded digest of the oracle's two-minute pass on the GPU box's host cores (tests/golden/cpu_oracle_proof_digests.json)
    assert hashlib.sha256(bytes(t["proof"])).hexdigest() == _cpu_oracle_digest("agg_k22_a3+1/evm/witness0")
    bf, n = sh.blinding_factors, 1 << sh.k
    host = dict(lookup_permuted=ctx.to_host(ctx.synth_fill(2 * (bf + 1), 11)), perm_z=ctx.to_host(ctx.synth_fill(sh.n_perm_sets * bf, 12)),
                lookup_z=ctx.to_host(ctx.synth_fill(bf, 13)), random_poly=ctx.to_host(ctx.synth_fill(n, 14)))
    t2 = gp.prove_native(w, transcript="evm", blinding=host, host_inputs=True)
    assert t2["proof"] != t["proof"] and verify_proof(gp, w, t2["proof"], "evm")
    gp.b.params.free()
    del gp, w


def test_survey_witness_mix_is_satisfiable(zk, oracle):
    """the SURVEY.md 8(d) value mix on the free witness cells (88-bit limbs, bits, uniform values: bench.py's informational
    `agg22_survey_witness` run) still gives a satisfiable instance: GPU proof == oracle proof, and it verifies"""
    from verify_util import verify_proof

    ffi, ctx = zk
    sh = pv.CircuitShape.agg(9)
    gp = pv.Prover(pv.GpuBackend(ctx, ffi), sh, satisfiable=True)
    cp = pv.Prover(OracleBackend(8), sh, satisfiable=True)
    w = gp.witness(0, dist="survey")
    for a, b in zip(w["advice"], cp.witness(0, dist="survey")["advice"]):
        assert (ctx.to_host(a) == b).all()
    t = gp.prove_native(w, transcript="evm")
    assert t["proof"] == cp.prove(cp.witness(0, dist="survey"), transcript="evm")["proof"]
    assert verify_proof(gp, w, t["proof"], "evm")


@pytest.mark.parametrize("k", [9, 13])
def test_proof_on_a_params_file_of_unknown_trapdoor_passes_the_pairing_check(zk, oracle, tmp_path, k):
    """The reference's acceptance criterion is the pairing (evm_verify, /root/reference/src/bin/cli.rs:524; src/tests/x509_aggregation.rs:110),
    and its SRS comes from a file (gen_srs, cli.rs:222,306).  A CHILD PROCESS draws a trapdoor from os.urandom, writes kzg_bn254_<k>.srs and exits
    without telling anyone; this process reads the file, proves on it, and the byte-driven verifier closes SHPLONK with
    e(L, g2) = e(h2, s_g2) over the file's two G2 points.  The synthetic SRS's trapdoor equation must fail on the same bytes."""
    import os
    import subprocess
    import sys

    from verify_util import srs_g2_from_params_file, verify_proof

    ffi, ctx = zk
    path = str(tmp_path / f"kzg_bn254_{k}.srs")
    code = ("import os, sys; sys.path.insert(0, %r)\n"
            "import halo2_zkcert_amd.ffi as ffi, halo2_zkcert_amd.prover as pv\n"
            "R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001\n"
            "ctx = ffi.Context(0); s = int.from_bytes(os.urandom(48), 'little') %% R\n"
            "ffi.ParamsKZG.setup(ctx, %d, pv.fr_from_int_host(s)).write(%r)\n" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), k, path))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip() == "", out.stderr[-2000:]
    g2 = srs_g2_from_params_file(path)
    assert g2[0] == __import__("pyref").G2_GEN and g2[1] is not None and g2[1] != g2[0]
    backend = pv.GpuBackend(ctx, ffi)
    backend.params_file = path
    sh = pv.CircuitShape.small(k)
    gp = pv.Prover(backend, sh, satisfiable=True)
    assert backend.params_source == path
    wit = gp.witness(0)
    for kind in ("poseidon", "evm"):
        proof = gp.prove_native(wit, transcript=kind)["proof"]
        assert verify_proof(gp, wit, proof, kind, srs_g2=g2)
        assert not verify_proof(gp, wit, proof, kind)                       # the public synthetic trapdoor is not this file's
        bad = bytearray(proof)
        bad[len(bad) // 2] ^= 1
        assert not verify_proof(gp, wit, bytes(bad), kind, srs_g2=g2)
    gp.release()
    backend.params.free()


@pytest.mark.parametrize("k", [6, 9, 12])
def test_two_phase_circuit_native_schedule_and_oracle_agree(zk, oracle, k):
    """Advice phases and a user challenge (zk_proving_key.advice_column_phase / challenge_phase, zk_proof_inputs.advice_phase; upstream's
    loop over `phases` in plonk/prover.rs [UPSTREAM-RECALL]): CircuitShape.two_phase — a column of phase 1 equal to challenge_0 * advice_0
    under its gate.  zkhip_create_proof_ex (the witness of phase 1 synthesised in the callback, device and host columns), the Python
    schedule on the GPU and the schedule on the oracle backend give the same proof BYTES under Poseidon and Keccak, and the byte-driven
    verifier — which reads the commitments phase by phase and squeezes the challenge in between — accepts them."""
    from verify_util import verify_proof

    ffi, ctx = zk
    sh = pv.CircuitShape.two_phase(k)
    gp = pv.Prover(pv.GpuBackend(ctx, ffi), sh, satisfiable=True)
    cp = pv.Prover(OracleBackend(8), sh, satisfiable=True)
    for kind in ("poseidon", "evm"):
        tc = cp.prove(cp.witness(0), transcript=kind)
        tg = gp.prove(gp.witness(0), transcript=kind)
        wn = gp.witness(0)
        tn = gp.prove_native(wn, transcript=kind)
        th = gp.prove_native(gp.witness(0), transcript=kind, host_inputs=True)
        assert tc["challenges"]["user"] == tg["challenges"]["user"] == tn["challenges"]["user"] and len(tn["challenges"]["user"]) == 1
        assert tn["proof"] == tc["proof"] and tg["proof"] == tc["proof"] and th["proof"] == tc["proof"]
        assert verify_proof(gp, wn, tn["proof"], kind)
    # the key says phase 1 but no callback is given: an argument error before anything enters the transcript
    import ctypes as C

    pk = gp._native_key()
    inp = ffi.ZkProofInputs()
    w = gp.witness(0)
    adv = (C.c_void_p * sh.n_advice)(*[c_.data_ptr() for c_ in w["advice"]])
    ins = (C.c_void_p * 1)(*[c_.data_ptr() for c_ in w["instance"]])
    inp.advice, inp.d_instance = C.cast(adv, C.c_void_p), C.cast(ins, C.c_void_p)
    nt = ffi.LibTranscript("poseidon")
    rc = ffi.lib().zkhip_create_proof_ex(ctx.h, C.byref(pk), C.byref(inp), nt.callbacks, None)
    assert rc == -1 and b"advice_phase is NULL" in ffi.lib().zkhip_last_error()
    gp.release()
    gp.b.params.free()


def test_hip_consumer_of_the_prover_vectors_on_a_self_made_file(zk, oracle, tmp_path, monkeypatch):
    """tests/test_reference_vectors.py::test_hip_create_proof_equals_upstreams_bytes — zkhip_create_proof_ex on an EXPLICIT circuit (key data,
    witness and every rng draw from the file, Prover.from_explicit + zk_blinding), bytes and challenges compared — driven by a file of the same
    schema written from the oracle (tests/refvec_util.py: self-made, pins nothing): the consumer that will meet upstream's file
    (integration/rust/refvec, /root/reference/Cargo.lock:1320-1322) runs on the GPU today, for small(6) and two_phase(6) under all three transcripts."""
    import importlib

    import halo2_zkcert_amd.formats as fm
    import refvec_util as ru
    from verify_util import verify_proof, vk_commitments

    ffi, ctx = zk
    doc = {"prover": ru.emit_prover_section(pv, oracle, ffi, fm, verify_proof, vk_commitments, tmp_path)}
    path = tmp_path / "reference_vectors.json"
    path.write_text(json.dumps(doc))
    mod = importlib.import_module("test_reference_vectors")
    monkeypatch.setattr(mod, "PATH", str(path))
    for which in ("small", "two_phase"):
        mod.test_hip_create_proof_equals_upstreams_bytes(zk, oracle, which)


def test_malformed_phases_are_rejected_before_any_work(zk, oracle):
    """zk_proving_key.advice_column_phase / challenge_phase (upstream: cs.advice_column_phase() / cs.challenge_phase(), handed out in order by
    ConstraintSystem): a phase without columns, a challenge after the last advice phase, or later phases without the advice_phase callback are
    ZKHIP_EINVAL with a message — not a NULL call or a silent single-phase proof."""
    ffi, ctx = zk
    sh = pv.CircuitShape.two_phase(6)
    p = pv.Prover(pv.GpuBackend(ctx, ffi), sh, satisfiable=True)
    w = p.witness(0)
    good = p.prove_native(w, transcript="poseidon")["proof"]
    sh.advice_phase[-1] = 2                     # phases 0 and 2, none of phase 1
    p._npk = None
    with pytest.raises(ffi.ZkhipError, match="a phase without columns"):
        p.prove_native(p.witness(0), transcript="poseidon")
    sh.advice_phase[-1] = 1
    sh.challenge_phase[0] = 2                   # a challenge of a phase after the last advice phase
    p._npk = None
    with pytest.raises(ffi.ZkhipError, match="the last advice phase is 1"):
        p.prove_native(p.witness(0), transcript="poseidon")
    sh.challenge_phase[0] = 0
    p._npk = None
    assert p.prove_native(p.witness(0), transcript="poseidon")["proof"] == good


def test_a_proof_four_times_the_baseline_size_verifies():
    """Maximum sizes: the aggregation-shaped circuit at k = 24 (2^24 rows, the quotient on 3 cosets of 2^24, transforms of 2^24, 138 GiB resident on the 288 GB device — four
    times BASELINE configs[3], /root/reference/src/bin/cli.rs:464-527; 0.49 s per proof) proved by one zkhip_create_proof_ex call under Keccak on a context of its own: the proof
    BYTES pass the byte-driven verifier, and the pageable-host-input path (worker-thread uploads of 4 x 512 MiB columns) gives the same bytes."""
    k = 24
    import torch

    import halo2_zkcert_amd.ffi as ffi
    from verify_util import verify_proof

    torch.cuda.empty_cache()      # (the session's other tests leave cached blocks behind: this one wants 140 GiB of the device's 288)
    ctx = ffi.Context(0)
    try:
        sh = pv.CircuitShape.agg(k, 3, 1)
        gp = pv.Prover(pv.GpuBackend(ctx, ffi), sh, satisfiable=True)
        w = gp.witness(0)
        t = gp.prove_native(w, transcript="evm")
        assert t["n_commitments"] == 16 and verify_proof(gp, w, t["proof"], "evm")
        assert gp.prove_native(w, transcript="evm", host_inputs="pageable")["proof"] == t["proof"]
        gp.release()
        gp.b.params.free()
        del gp, w
    finally:
        torch.cuda.empty_cache()
        ctx.close()
