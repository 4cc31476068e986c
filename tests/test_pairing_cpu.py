"""The oracle's BN254 optimal-ate pairing (oracle/pyref.py) and the verifier closed by it: the reference's acceptance
criterion is a pairing check (evm_verify, /root/reference/src/bin/cli.rs:524, src/tests/x509_aggregation.rs:110), so a proof made on a real
SRS — trapdoor unknown — can be checked.  Anchors: EIP-197's G2 generator (on the twist, order r), a PUBLISHED alt_bn128 pairing test input
(go-ethereum's bn256Pairing "jeff1", also in the EIP-197 state tests: two pairs whose product is one; its G1 points have no known relation to
the generator, and its first G2 point is not the generator), bilinearity and non-degeneracy."""
import numpy as np
import pytest

import pyref as P

JEFF1 = """1c76476f4def4bb94541d57ebba1193381ffa7aa76ada664dd31c16024c43f59 3034dd2920f673e204fee2811c678745fc819b55d3e9d294e45c9b03a76aef41
209dd15ebff5d46c4bd888e51a93cf99a7329636c63514396b4a452003a35bf7 04bf11ca01483bfa8b34b43561848d28905960114c8ac04049af4b6315a41678
2bb8324af6cfc93537a2ad1a445cfd0ca2a71acd7ac41fadbf933c2a51be344d 120a2a4cf30c1bf9845f20c6fe39e07ea2cce61f0c9bb048165fe5e4de877550
111e129f1cf1097710d41c4ac70fcdfa5ba2023c6ff1cbeac322de49d1b6df7c 2032c61a830e3c17286de9462bf242fca2883585b93870a73853face6a6bf411
198e9393920d483a7260bfb731fb5d25f1aa493335a9e71297e485b7aef312c2 1800deef121f1e76426a00665e5c4479674322d4f75edadd46debd5cd992f6ed
090689d0585ff075ec9e99ad690c3395bc4b313370b38ef355acdadcd122975b 12c85ea5db8c6deb4aab71808dcb408fe3d1e7690c43d37b4ce6cc0166fa7daa"""


def _eip197_pairs(words):
    """EIP-197 input: per pair G1 (x, y), then G2 as (x imaginary, x real, y imaginary, y real), 32-byte big-endian words"""
    w = [int(x, 16) for x in words.split()]
    return [((w[i], w[i + 1]), ((w[i + 3], w[i + 2]), (w[i + 5], w[i + 4]))) for i in range(0, len(w), 6)]


def test_g2_generator_and_tower():
    assert P.g2_on_curve(P.G2_GEN) and P.g2_mul(P.R, P.G2_GEN) is None and P.g2_mul(P.R - 1, P.G2_GEN) == P.g2_neg(P.G2_GEN)
    assert P.f2_mul(P.XI, P.f2_inv(P.XI)) == P.F2_ONE
    # v^3 = xi and w^2 = v in the tower as multiplied
    v = (P.F2_ZERO, P.F2_ONE, P.F2_ZERO)
    assert P.f6_mul(P.f6_mul(v, v), v) == (P.XI, P.F2_ZERO, P.F2_ZERO)
    w = (P.F6_ZERO, P.F6_ONE)
    assert P.f12_mul(w, w) == (v, P.F6_ZERO)


def test_published_eip197_vector():
    pairs = _eip197_pairs(JEFF1)
    assert all(P.on_curve(p1) and P.g2_on_curve(q2) for p1, q2 in pairs)
    assert pairs[1][1] == P.G2_GEN and pairs[0][1] != P.G2_GEN
    assert P.pairing_check(pairs)
    # any change breaks it: the second G1 point negated, or the pairs' G2 points swapped
    (a1, a2), (b1, b2) = pairs
    assert not P.pairing_check([(a1, a2), ((b1[0], P.P - b1[1]), b2)])
    assert not P.pairing_check([(a1, b2), (b1, a2)])


def test_bilinear_and_non_degenerate():
    e = P.pairing((1, 2), P.G2_GEN)
    assert e != P.F12_ONE and P.f12_pow(e, P.R) == P.F12_ONE
    a, b = 0x1D5C0FFEE12345, 0xFEEDFACE98765
    pa = P.to_affine(P.scalar_mul(a, P.from_affine((1, 2))))
    assert P.pairing(pa, P.g2_mul(b, P.G2_GEN)) == P.f12_pow(e, a * b % P.R)
    # the precompile's product form, and identity handling
    assert P.pairing_check([(pa, P.G2_GEN), ((1, P.P - 2), P.g2_mul(a, P.G2_GEN))])
    assert P.pairing_check([((0, 0), P.G2_GEN), (pa, None)])


def test_params_file_g2_bytes_parse_to_the_generator_and_its_multiple():
    import halo2_zkcert_amd.ffi as ffi

    s = 0x5EED5EED5EED
    raw = ffi._g2_setup_bytes(s)
    assert len(raw) == 256 and P.g2_from_raw_bytes(raw[:128]) == P.G2_GEN and P.g2_from_raw_bytes(raw[128:]) == P.g2_mul(s, P.G2_GEN)
    bad = bytearray(raw[:128])
    bad[0] ^= 1
    with pytest.raises(ValueError):
        P.g2_from_raw_bytes(bytes(bad))


@pytest.mark.parametrize("kind", ["poseidon", "evm"])
def test_pairing_verifier_agrees_with_the_trapdoor_verifier(oracle, kind):
    """the same proof bytes under both closings of SHPLONK: accepted by both; a tampered proof and a wrong [s]_2 rejected"""
    import halo2_zkcert_amd.prover as pv
    from oracle_backend import OracleBackend
    from verify_util import verify_proof

    s = 0x1D5C0FFEE
    sh = pv.CircuitShape.small(5)
    p = pv.Prover(OracleBackend(2), sh, srs_trapdoor=s, satisfiable=True)
    wit = p.witness(0)
    proof = p.prove(wit, transcript=kind)["proof"]
    g2 = (P.G2_GEN, P.g2_mul(s, P.G2_GEN))
    assert verify_proof(p, wit, proof, kind) and verify_proof(p, wit, proof, kind, srs_g2=g2)
    assert not verify_proof(p, wit, proof, kind, srs_g2=(P.G2_GEN, P.g2_mul(s + 1, P.G2_GEN)))
    bad = bytearray(proof)
    bad[len(bad) - 40] ^= 1          # inside the last evaluation / opening point
    assert not verify_proof(p, wit, bytes(bad), kind, srs_g2=g2)
