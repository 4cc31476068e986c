"""Reference-HELD vectors: tests/golden/reference_vectors.json is printed by integration/rust/refvec from the UNPATCHED crates the
reference pins (halo2curves e185711, halo2_proofs 4b42325, snark-verifier-sdk 7011e8c) on a machine that has Rust.  No such machine was
available to the build, so the file is absent and these tests skip; the day it is dropped in, they pin the oracle, the library's host
code (encodings, transcripts) and — under -m gpu — the HIP kernels against outputs of the reference's own code."""
import json
import os

import numpy as np
import pytest

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors.json")
# Reported as XFAIL (an expected, named gap in the summary line), not as a silent skip, while the file is absent: "parity unpinned against upstream" stays visible in
# every test run.  With the file present the marker does nothing (run = the condition is false).
pytestmark = pytest.mark.xfail(not os.path.exists(PATH), run=False, strict=False,
                               reason="PARITY UNPINNED against upstream: tests/golden/reference_vectors.json has not been generated (integration/rust/refvec needs a machine with Rust)")
H = lambda s: int(s, 16)


def _load():
    with open(PATH) as f:
        return json.load(f)


def _synth(zo, seed, n):
    return zo.synth_raw253(seed, n)


def test_constants_and_encodings(oracle):
    import pyref as P

    import halo2_zkcert_amd.ffi as ffi

    zo = oracle
    v = _load()["constants"]
    assert H(v["zeta"]) == P.ZETA and H(v["delta"]) == P.DELTA and H(v["root_of_unity"]) == P.ROOT_OF_UNITY and v["s"] == P.S
    g = zo.affine_from_ints([(1, 2)])[0]
    assert ffi.g1_to_bytes(g).hex() == v["generator_compressed"]
    assert ffi.g1_to_bytes(np.zeros(8, dtype=np.uint64)).hex() == v["identity_compressed"]
    neg = zo.affine_from_ints([(1, P.P - 2)])[0]
    assert ffi.g1_to_bytes(neg).hex() == v["neg_g_compressed"]
    two = zo.g1_mul_gen(zo.fr_from_int(2))
    assert ffi.g1_to_bytes(two).hex() == v["two_g"]["compressed"] and zo.g1_to_bytes(two).hex() == v["two_g"]["compressed"]


def test_transcripts(oracle):
    import halo2_zkcert_amd.ffi as ffi
    import halo2_zkcert_amd.prover as pv

    zo = oracle
    v = _load()["transcripts"]
    p1 = zo.g1_mul_gen(zo.fr_from_int(5))
    for kind in ("blake2b", "poseidon", "evm"):
        t = ffi.LibTranscript(kind)
        t.common_scalar(zo.fr_from_int(7))
        t.write_point(p1)
        t.write_scalar(zo.fr_from_int(0x1234567890ABCDEF))
        c1, c2 = pv.from_mont_host(t.squeeze_limbs()), pv.from_mont_host(t.squeeze_limbs())
        assert (c1, c2) == (H(v[kind]["c1"]), H(v[kind]["c2"])), kind
        assert t.proof().hex() == v[kind]["proof"], kind


def test_oracle_msm_fft_domain(oracle):
    zo = oracle
    v = _load()
    for c in v["msm"]:
        n = 1 << c["k"]
        mono, _ = zo.kzg_setup_scalars(c["k"], zo.fr_from_int(H(c["srs_trapdoor"])))
        bases = zo.fixed_base_mul(mono, 4)
        s = zo.g1_to_affine(zo.best_multiexp(_synth(zo, c["seed"], n), bases, 4))
        assert zo.g1_to_bytes(s).hex() == c["sum"]["compressed"]
    for c in v["fft"]:
        n = 1 << c["k"]
        a = zo.best_fft(_synth(zo, c["seed"], n), zo.fr_from_int(H(c["omega"])), c["k"], 4)
        ints = zo.fr_arr_to_ints(a)
        assert (ints[0], ints[1], ints[-1]) == (H(c["first"]), H(c["second"]), H(c["last"]))
        if c["all"]:
            assert ints == [H(x) for x in c["all"]]
    d = v["domain"]
    dom = zo.Domain(d["j"], d["k"])
    assert dom.extended_k == d["extended_k"]
    coeff = dom.lagrange_to_coeff(_synth(zo, d["seed"], 1 << d["k"]), 4)
    assert zo.fr_arr_to_ints(coeff) == [H(x) for x in d["coeff"]]
    assert zo.fr_arr_to_ints(dom.coeff_to_extended(coeff, 4)) == [H(x) for x in d["extended"]]


@pytest.mark.gpu
def test_hip_msm_fft_domain(zk, oracle):
    ffi, ctx = zk
    zo = oracle
    v = _load()
    for c in v["msm"]:
        n = 1 << c["k"]
        p = ffi.ParamsKZG.setup(ctx, c["k"], zo.fr_from_int(H(c["srs_trapdoor"])))
        s = ffi.g1_to_affine(p.commit(_synth(zo, c["seed"], n)))
        assert ffi.g1_to_bytes(s).hex() == c["sum"]["compressed"]
        p.free()
    for c in v["fft"]:
        a = ctx.best_fft(_synth(zo, c["seed"], 1 << c["k"]), zo.fr_from_int(H(c["omega"])), c["k"])
        ints = zo.fr_arr_to_ints(a)
        assert (ints[0], ints[1], ints[-1]) == (H(c["first"]), H(c["second"]), H(c["last"]))
    d = v["domain"]
    dom = ffi.EvaluationDomain(ctx, d["j"], d["k"])
    coeff = dom.lagrange_to_coeff(_synth(zo, d["seed"], 1 << d["k"]))
    assert zo.fr_arr_to_ints(coeff) == [H(x) for x in d["coeff"]]
    assert zo.fr_arr_to_ints(dom.coeff_to_extended(coeff)) == [H(x) for x in d["extended"]]


# ------------------------------------------------------------------------------------------------------------------------ the `prover` section
# Whole create_proof runs of the UNPATCHED crates on explicit circuits (integration/rust/refvec/src/prover_vectors.rs): the rows of
# SURVEY.md §8 with the most recall in them — the proving schedule (a8), through it evaluate_h's quotient (a7) and the on-disk formats (f3).
# The reference's own callers of that code: /root/reference/src/helpers.rs:213,233, src/bin/cli.rs:312,478-483,519.
def _prove_like_the_file(backend, doc, zo, python_schedule):
    import halo2_zkcert_amd.prover as pv
    import refvec_util as ru

    ru.check_constraint_system(pv, doc)
    sh = ru.shape_of(pv, doc["circuit"])
    out = {}
    for kind in ("blake2b", "poseidon", "evm"):
        ref = doc["proofs"][kind]
        assert ref["verified"] is True, "upstream's own verifier rejected upstream's proof: the Rust circuit is not what it says"
        model = ru.pick_model(sh, ref["rng_u64_drawn"])
        roles = ru.draw_roles(sh, model)
        p, wit, blinding = ru.prover_and_witness(pv, zo, backend, doc, roles)
        t = p.prove(wit, transcript=kind, blinding=blinding) if python_schedule else p.prove_native(wit, transcript=kind, blinding=blinding)
        order = [("user", i) for i in range(len(sh.challenge_phase))] + ["theta", "beta", "gamma", "y", "x", "shplonk_y", "shplonk_v", "shplonk_u"]
        chs = [t["challenges"]["user"][tag[1]] if isinstance(tag, tuple) else t["challenges"][tag] for tag in order]
        want = [H(x) for x in ref["challenges"]]
        first_bad = next((i for i, (a, b) in enumerate(zip(chs, want)) if a != b), None)
        assert len(chs) == len(want) and first_bad is None, f"{kind} ({model}): challenge #{first_bad} ({order[first_bad] if first_bad is not None else '-'}) differs from upstream's"
        assert bytes(t["proof"]).hex() == ref["proof"], f"{kind} ({model}): proof bytes differ from upstream's"
        out[kind] = (p, model)
    return out


@pytest.mark.parametrize("which", ["small", "two_phase"])
def test_oracle_create_proof_equals_upstreams_bytes(oracle, which):
    """the CPU oracle's create_proof on upstream's circuit, key data, witness and rng draws == upstream's proof bytes and challenges, under all
    three transcripts; the verifying key's fixed / permutation commitments are upstream's"""
    from oracle_backend import OracleBackend
    from verify_util import vk_commitments

    zo = oracle
    doc = _load()["prover"][which]
    got = _prove_like_the_file(OracleBackend(2), doc, zo, python_schedule=True)
    p = got["poseidon"][0]
    fixed, sigma = vk_commitments(p)
    pt = lambda xy: zo.g1_to_bytes(zo.affine_from_ints([xy])[0]).hex()
    assert [pt(x) for x in fixed] == doc["fixed_commitments"] and [pt(x) for x in sigma] == doc["permutation_commitments"]


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["small", "two_phase"])
def test_hip_create_proof_equals_upstreams_bytes(zk, oracle, which):
    """zkhip_create_proof_ex (one call, the caller's rng draws handed over as zk_blinding) == upstream's proof bytes and challenges"""
    import halo2_zkcert_amd.prover as pv

    ffi, ctx = zk
    _prove_like_the_file(pv.GpuBackend(ctx, ffi), _load()["prover"][which], oracle, python_schedule=False)


def test_files_written_by_upstream_parse_and_round_trip(oracle, tmp_path):
    """ParamsKZG::write, ProvingKey::write(RawBytesUnchecked) and bincode(Snark) as upstream wrote them: formats.py reads them, finds upstream's
    circuit in them, and writes the same bytes back"""
    import halo2_zkcert_amd.ffi as ffi
    import halo2_zkcert_amd.formats as fm
    import halo2_zkcert_amd.prover as pv
    import refvec_util as ru
    from oracle_backend import OracleBackend

    zo = oracle
    v = _load()["prover"]
    files, doc = v["files"], v["small"]
    s = H(v["srs_trapdoor"])
    # --- SRS
    raw = bytes.fromhex(files["params"])
    pf = fm.ParamsFile.parse(raw)
    mono, lag = zo.kzg_setup_scalars(pf.k, zo.fr_from_int(s))
    assert pf.k == doc["circuit"]["k"] and (pf.g == zo.fixed_base_mul(mono, 2)).all() and (pf.g_lagrange == zo.fixed_base_mul(lag, 2)).all()
    assert pf.g2_bytes == ffi._g2_setup_bytes(s) and pf.to_bytes() == raw
    # --- proving key: upstream's columns are the circuit's, its sigma / coefficient forms are what this repository's keygen derives
    sh = ru.shape_of(pv, doc["circuit"])
    p, _, _ = ru.prover_and_witness(pv, zo, OracleBackend(2), doc, ru.draw_roles(sh, "kzg"))
    path = tmp_path / "upstream.pk"
    path.write_bytes(bytes.fromhex(files["pk"]))
    pk = fm.ProvingKeyFile.read(path, n_perm_columns=len(sh.perm_columns), n_selectors=files["pk_n_selectors"])
    assert pk.k == sh.k and len(pk.fixed_values) == sh.n_fixed
    for mine, theirs in zip(p.fixed_lagrange + p.sigma_lagrange + p.fixed_coeff + p.sigma_coeff,
                            list(pk.fixed_values) + list(pk.permutations) + list(pk.fixed_polys) + list(pk.permutation_polys)):
        assert (np.asarray(mine) == np.asarray(theirs)).all()
    ext = p._extended_key()
    for mine, theirs in zip(ext["fixed"] + ext["sigma"] + [ext["l0"], ext["l_last"], ext["l_active"]],
                            list(pk.fixed_cosets) + list(pk.permutation_cosets) + [pk.l0, pk.l_last, pk.l_active_row]):
        assert (np.asarray(mine) == np.asarray(theirs)).all()
    pt = lambda xy: zo.g1_to_bytes(np.asarray(xy, dtype=np.uint64)).hex()
    assert [pt(x) for x in pk.fixed_commitments] == doc["fixed_commitments"] and [pt(x) for x in pk.permutation_commitments] == doc["permutation_commitments"]
    back = tmp_path / "back.pk"
    pk.write(back)
    assert back.read_bytes() == path.read_bytes()
    # --- snark
    spath = tmp_path / "upstream.snark"
    spath.write_bytes(bytes.fromhex(files["snark"]))
    sn = fm.SnarkFile.read(spath, protocol_len=files["snark_protocol_len"])
    assert sn.proof.hex() == files["snark_proof"] and sn.instances == [[H(x) for x in c] for c in files["snark_instances"]]
    found = fm.SnarkFile.read(spath)                      # the suffix scan locates the same split without being told
    assert len(found.protocol) == files["snark_protocol_len"] and found.proof == sn.proof
    assert sn.protocol + sn.tail_bytes() == spath.read_bytes()
