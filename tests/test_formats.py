"""CPU: the on-disk artefacts of the reference's CLI steps as restated in halo2_zkcert_amd/formats.py (round trips; no reference-made
file exists here, see the module's header)."""
import json

import numpy as np
import pytest

import halo2_zkcert_amd.formats as fm
import halo2_zkcert_amd.prover as pv
from oracle_backend import OracleBackend


def test_break_points_json(tmp_path):
    bp = [[262133, 262134, 262130], []]
    p = tmp_path / "agg_break_points.json"
    fm.write_break_points(p, bp)
    assert p.read_text() == "[[262133,262134,262130],[]]"            # serde_json::to_string's compact form
    assert fm.read_break_points(p, k=22) == bp
    assert fm.advice_columns_from_break_points(bp) == [4, 1]
    p.write_text('{"a": 1}')
    with pytest.raises(ValueError):
        fm.read_break_points(p)
    p.write_text("[[5, -1]]")
    with pytest.raises(ValueError):
        fm.read_break_points(p)
    p.write_text("[[1048576]]")
    with pytest.raises(ValueError):
        fm.read_break_points(p, k=20)


def test_proving_key_round_trip(oracle, tmp_path):
    """ProvingKey::write / ::read (RawBytesUnchecked layout): a keygen-shaped key from the oracle backend goes to disk and comes back
    array for array; a wrong column count is detected by the trailing-bytes check; the file size is the layout's."""
    zo = oracle
    sh = pv.CircuitShape.small(5)
    p = pv.Prover(OracleBackend(2), sh, satisfiable=True)
    fixed_c = [c[0] for c in p.b.commit(p.fixed_coeff, lagrange=False)]
    sigma_c = [c[0] for c in p.b.commit(p.sigma_coeff, lagrange=False)]
    n, en = 1 << sh.k, p.dom.extended_n
    sel = [np.arange(n) % 4 == 0, np.arange(n) % 3 == 1]
    pk = fm.ProvingKeyFile.from_prover(p, fixed_c, sigma_c, selectors=sel)
    path = tmp_path / "rsa_1.pk"
    pk.write(path)
    nf, npm = len(p.fixed_lagrange), len(p.sigma_lagrange)
    poly = lambda m: 4 + 32 * m
    expect = 4 + 4 + 64 * nf + 64 * npm + 2 * (n // 8) + 3 * poly(en) + (4 + nf * poly(n)) * 2 + (4 + nf * poly(en)) + (4 + npm * poly(n)) * 2 + (4 + npm * poly(en))
    assert path.stat().st_size == expect
    back = fm.ProvingKeyFile.read(path, n_perm_columns=npm, n_selectors=2)
    assert back.k == sh.k and back.extended_k == p.dom.extended_k
    assert (back.fixed_commitments == np.stack(fixed_c)).all() and (back.permutation_commitments == np.stack(sigma_c)).all()
    assert all((a == b).all() for a, b in zip(back.selectors, sel))
    for a, b in zip(back.fixed_values + back.fixed_polys + back.fixed_cosets + back.permutations + back.permutation_polys + back.permutation_cosets,
                    p.fixed_lagrange + p.fixed_coeff + p.fixed_cosets + p.sigma_lagrange + p.sigma_coeff + p.sigma_cosets):
        assert (a == b).all()
    assert (back.l0 == p.l0).all() and (back.l_last == p.l_last).all() and (back.l_active_row == p.l_active).all()
    with pytest.raises(ValueError):
        fm.ProvingKeyFile.read(path, n_perm_columns=npm + 1, n_selectors=2)
    with pytest.raises(ValueError):
        fm.ProvingKeyFile.read(path, n_perm_columns=npm, n_selectors=1)
    # a generator in Montgomery form: the first fixed commitment decodes to a curve point
    x, y = zo.affine_to_ints(back.fixed_commitments[:1])[0]
    assert (y * y - x * x * x - 3) % 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47 == 0


def test_snark_file_round_trip(oracle, tmp_path):
    """bincode(Snark): instances and proof are recovered from a file whose protocol part is opaque, with the protocol's length given
    and by the suffix scan; the proof bytes are a real proof of the schedule (Poseidon transcript, as gen_snark_shplonk writes)."""
    sh = pv.CircuitShape.small(5)
    p = pv.Prover(OracleBackend(2), sh, satisfiable=True)
    w = p.witness(0)
    proof = p.prove(w, transcript="poseidon")["proof"]
    inst = [[pv.from_mont_host(v) for v in col] for col in w["instance_values"]]
    protocol = bytes((i * 37 + 11) & 0xFF for i in range(777))
    s = fm.SnarkFile(protocol, inst, proof)
    path = tmp_path / "rsa_1.proof"
    s.write(path)
    a = fm.SnarkFile.read(path, protocol_len=len(protocol))
    b = fm.SnarkFile.read(path)
    for got in (a, b):
        assert got.protocol == protocol and got.instances == inst and got.proof == proof
    with pytest.raises(ValueError):
        fm.SnarkFile.read(path, protocol_len=len(protocol) + 1)


def test_prover_from_proving_key_file(oracle, tmp_path):
    """read_pk -> create_proof (/root/reference/src/bin/cli.rs:312,320): a Prover built from a ProvingKeyFile generates nothing of the key
    and, with the witness that was saved beside it, reproduces the proof bytes of the prover that wrote the file — on the extended-domain
    path, which reads the file's fixed_cosets / permutation cosets / l-polynomials."""
    sh = pv.CircuitShape.small(6)
    p1 = pv.Prover(OracleBackend(2), sh, satisfiable=True)
    w1 = p1.witness(2)
    ref = p1.prove(w1, transcript="poseidon")["proof"]
    fixed_c = [c[0] for c in p1.b.commit(p1.fixed_coeff, lagrange=False)]
    sigma_c = [c[0] for c in p1.b.commit(p1.sigma_coeff, lagrange=False)]
    fm.ProvingKeyFile.from_prover(p1, fixed_c, sigma_c).write(tmp_path / "k.pk")
    p1.save_witness(w1, tmp_path / "w.npz")
    kf = fm.ProvingKeyFile.read(tmp_path / "k.pk", n_perm_columns=len(sh.perm_columns), n_selectors=0)
    p2 = pv.Prover(OracleBackend(2), sh, key_file=kf)
    assert p2.key_source == "file" and not hasattr(p2, "_value_src")
    w2 = p2.load_witness(tmp_path / "w.npz")
    assert p2.prove(w2, transcript="poseidon")["proof"] == ref
    assert p2._ext is not None and (p2.l0 == kf.l0).all()                       # the extended forms came from the file
    with pytest.raises(ValueError):
        pv.Prover(OracleBackend(2), pv.CircuitShape.small(7), key_file=kf)
