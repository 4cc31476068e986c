"""The reference-vector pin must be ONE command away: integration/rust/refvec/src/main.rs (run on a machine with Rust against the
unpatched pinned crates, /root/reference/Cargo.lock:1320-1322,1359-1361,2714-2716) has to emit every key tests/test_reference_vectors.py
reads, and those tests themselves have to be runnable.  Checked here without Rust: (1) statically — every JSON key the tests index
appears as a key in the json!{} literals of main.rs; (2) dynamically — a file of the SAME schema written from this repo's own oracle
(labelled SELF-MADE: it pins nothing) drives the three CPU tests of test_reference_vectors.py to green, so a typo in a key, an API
misuse or a wrong hex convention in the consumer cannot be what fails on the day the real file arrives."""
import importlib
import json
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAIN_RS = os.path.join(ROOT, "integration", "rust", "refvec", "src", "main.rs")
PROVER_RS = os.path.join(ROOT, "integration", "rust", "refvec", "src", "prover_vectors.rs")      # the `prover` section (round 5)
UTIL = os.path.join(ROOT, "tests", "refvec_util.py")                                             # ... and its consumer-side helper


def _emitted_keys():
    return set(re.findall(r'"([a-z_0-9]+)"\s*:', open(MAIN_RS).read() + open(PROVER_RS).read()))
CONSUMER = os.path.join(ROOT, "tests", "test_reference_vectors.py")


def test_refvec_emits_every_key_the_tests_read():
    rs = open(MAIN_RS).read() + open(PROVER_RS).read()
    emitted = _emitted_keys()
    src = open(CONSUMER).read() + open(UTIL).read()
    read = set(re.findall(r'\b(?:v|c|d|cs|doc|ref|files|circuit|_load\(\))\["([a-z_0-9]+)"\]', src)) | set(re.findall(r'\]\["([a-z_0-9]+)"\]', src))
    read |= set(re.findall(r'\b(?:doc|files|v)\.get\("([a-z_0-9]+)"', src))
    read |= {"blake2b", "poseidon", "evm", "small", "two_phase"}          # for kind in (...): v[kind]; parametrised sections
    read -= {"advice_for_phase", "user"}                                    # keys of this repository's own witness / trace dicts, not of the file
    assert len(read) > 50, read
    missing = sorted(read - emitted)
    assert not missing, f"tests/test_reference_vectors.py reads keys that refvec/src/main.rs never emits: {missing}"
    # the pins the program is to be built against are the reference's
    lock = os.path.join(ROOT, "integration", "rust", "refvec", "Cargo.toml")
    toml = open(lock).read()
    for rev in ("e185711", "4b42325", "7011e8c"):
        assert rev in toml or rev in rs, rev


def _self_made(zo, P, ffi, pv):
    """the schema of refvec's output, filled from this repo's oracle (NOT a pin)"""
    hx = lambda x: format(x, "064x")
    g = zo.affine_from_ints([(1, 2)])[0]
    two = zo.g1_mul_gen(zo.fr_from_int(2))
    constants = {"zeta": hx(P.ZETA), "delta": hx(P.DELTA), "root_of_unity": hx(P.ROOT_OF_UNITY), "s": P.S,
                 "generator_compressed": ffi.g1_to_bytes(g).hex(), "identity_compressed": ffi.g1_to_bytes(np.zeros(8, dtype=np.uint64)).hex(),
                 "neg_g_compressed": ffi.g1_to_bytes(zo.affine_from_ints([(1, P.P - 2)])[0]).hex(), "two_g": {"compressed": ffi.g1_to_bytes(two).hex()}}
    msm = []
    for k, seed in ((8, 11), (12, 12)):
        mono, _ = zo.kzg_setup_scalars(k, zo.fr_from_int(0x1D5C0FFEE))
        s = zo.g1_to_affine(zo.best_multiexp(zo.synth_raw253(seed, 1 << k), zo.fixed_base_mul(mono, 4), 4))
        msm.append({"k": k, "seed": seed, "srs_trapdoor": "1d5c0ffee", "sum": {"compressed": zo.g1_to_bytes(s).hex()}})
    fft = []
    for k, seed in ((4, 21), (10, 22)):
        omega = pow(P.ROOT_OF_UNITY, 1 << (P.S - k), P.R)
        a = zo.fr_arr_to_ints(zo.best_fft(zo.synth_raw253(seed, 1 << k), zo.fr_from_int(omega), k, 4))
        fft.append({"k": k, "seed": seed, "omega": hx(omega), "first": hx(a[0]), "second": hx(a[1]), "last": hx(a[-1]), "all": [hx(x) for x in a] if k <= 4 else []})
    dom = zo.Domain(4, 6)
    coeff = dom.lagrange_to_coeff(zo.synth_raw253(31, 64), 4)
    domain = {"j": 4, "k": 6, "extended_k": dom.extended_k, "seed": 31, "coeff": [hx(x) for x in zo.fr_arr_to_ints(coeff)],
              "extended": [hx(x) for x in zo.fr_arr_to_ints(dom.coeff_to_extended(coeff, 4))]}
    tr = {}
    p1 = zo.g1_mul_gen(zo.fr_from_int(5))
    for kind in ("blake2b", "poseidon", "evm"):
        t = ffi.LibTranscript(kind)
        t.common_scalar(zo.fr_from_int(7))
        t.write_point(p1)
        t.write_scalar(zo.fr_from_int(0x1234567890ABCDEF))
        c1, c2 = pv.from_mont_host(t.squeeze_limbs()), pv.from_mont_host(t.squeeze_limbs())
        tr[kind] = {"c1": hx(c1), "c2": hx(c2), "proof": t.proof().hex()}
    return {"source": "SELF-MADE from oracle/ (schema check only; pins nothing)", "constants": constants, "msm": msm, "fft": fft, "domain": domain, "transcripts": tr}


def test_the_explicit_circuit_is_satisfiable_and_matches_the_shape():
    """refvec_util.build_circuit (the restatement of prover_vectors.rs ShapeCircuit::build): every gate, copy and lookup holds on the usable
    rows — upstream's verifier must accept the proof the Rust side makes from the same values"""
    import refvec_util as ru

    for two_phase in (False, True):
        c = ru.build_circuit(two_phase)
        H = ru.H
        fixed, adv, inst = [[H(x) for x in col] for col in c["fixed"]], [[H(x) for x in col] for col in c["advice"]], [H(x) for x in c["instance"][0]]
        u = c["usable_rows"]
        assert u == 57 and all(len(col) == u for col in fixed + adv)
        for col in (0, 1):
            for r in range(u - 3):
                assert fixed[col][r] * (adv[col][r] + adv[col][r + 1] * adv[col][r + 2] - adv[col][r + 3]) % ru.R == 0
            assert all(fixed[col][r] == 0 for r in range(u - 3, u))          # no gate reaches into the blinding rows
        assert set(adv[2]) <= set(fixed[3])
        n_adv = 4 if two_phase else 3
        cell = lambda pc, r: adv[pc][r] if pc < 3 else (None if pc < n_adv else (fixed[2][r] if pc == n_adv else inst[r]))
        cells = [(a, b) for a, ra, b, rb in c["copies"] for a, b in [((a, ra), (b, rb))]]
        assert len(set(x for pair in cells for x in pair)) == 2 * len(cells)          # disjoint two-cycles
        for (a, ra), (b, rb) in cells:
            assert cell(a, ra) == cell(b, rb)


def test_consumer_runs_on_a_file_of_that_schema(oracle, tmp_path, monkeypatch):
    import pyref as P

    import __graft_entry__ as g

    g.build()
    import halo2_zkcert_amd.ffi as ffi
    import halo2_zkcert_amd.prover as pv

    import halo2_zkcert_amd.formats as fm
    import refvec_util as ru
    from verify_util import verify_proof, vk_commitments

    doc = _self_made(oracle, P, ffi, pv)
    doc["prover"] = ru.emit_prover_section(pv, oracle, ffi, fm, verify_proof, vk_commitments, tmp_path)
    assert all(doc["prover"][w]["proofs"][kind]["verified"] for w in ("small", "two_phase") for kind in ("blake2b", "poseidon", "evm"))
    # the self-made file has exactly the keys refvec emits (nested), no more, no fewer
    emitted = _emitted_keys()

    def keys(o):
        if isinstance(o, dict):
            for k_, v_ in o.items():
                yield k_
                yield from keys(v_)
        elif isinstance(o, list):
            for v_ in o:
                yield from keys(v_)
    assert set(keys(doc)) <= emitted, sorted(set(keys(doc)) - emitted)
    path = tmp_path / "reference_vectors.json"
    path.write_text(json.dumps(doc))
    mod = importlib.import_module("test_reference_vectors")
    monkeypatch.setattr(mod, "PATH", str(path))
    mod.test_constants_and_encodings(oracle)
    mod.test_transcripts(oracle)
    mod.test_oracle_msm_fft_domain(oracle)
    for which in ("small", "two_phase"):
        mod.test_oracle_create_proof_equals_upstreams_bytes(oracle, which)
    mod.test_files_written_by_upstream_parse_and_round_trip(oracle, tmp_path)
    # a stream with upstream's extra Blind draws is recognised by its count and mapped accordingly; any other count is reported, not guessed
    sh = ru.shape_of(pv, doc["prover"]["small"]["circuit"])
    assert ru.pick_model(sh, 8 * ru.expected_fr_draws(sh, "blinds")) == "blinds" and ru.pick_model(sh, 8 * ru.expected_fr_draws(sh, "kzg")) == "kzg"
    assert ru.draw_roles(sh, "blinds")["u64_drawn"] == 8 * ru.expected_fr_draws(sh, "blinds") and ru.draw_roles(sh, "blinds")["random_poly"] != ru.draw_roles(sh, "kzg")["random_poly"]
    with pytest.raises(AssertionError, match="upstream's order is another one"):
        ru.pick_model(sh, 8 * ru.expected_fr_draws(sh, "kzg") + 8)
