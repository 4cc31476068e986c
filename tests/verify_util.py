"""Runs oracle/pyref.py's algebraic PLONK + SHPLONK verifier over a Prover trace or over proof BYTES (TEST INFRASTRUCTURE)."""
import numpy as np

import pyref as P
import zkoracle_py as zo


def _pts(arrs):
    return [zo.affine_to_ints(np.asarray(a, dtype=np.uint64).reshape(1, 8))[0] for a in arrs]


def _vk(sh):
    return dict(k=sh.k, degree=sh.degree, blinding_factors=sh.blinding_factors, gates=sh.gates, lookups=sh.lookups,
                perm_columns=sh.perm_columns, n_advice=sh.n_advice, advice_phase=list(sh.advice_phase), challenge_phase=list(sh.challenge_phase))


def _queries(sh):
    aq = [(c, r) for kind, c, r in sh.queries() if kind == "advice"]
    fq = [(c, r) for kind, c, r in sh.queries() if kind == "fixed"]
    return aq, fq


def vk_commitments(prover, oracle_side=False, srs_trapdoor=0x1D5C0FFEE, threads=8):
    """the verifying key's fixed / sigma commitments (cached).  Default: committed with the prover's own backend, as keygen would.
    oracle_side=True: committed by the CPU ORACLE (its own SRS from the trapdoor, its own MSM) from host copies of the key's polynomials —
    the verifier's inputs then owe nothing to the GPU (at k = 17 the verifier was checking GPU commitments with GPU-made
    vk points)."""
    key = "_vk_coms_oracle" if oracle_side else "_vk_coms"
    if getattr(prover, key, None) is None:
        b = prover.b
        if oracle_side:
            mono, _ = zo.kzg_setup_scalars(prover.shape.k, zo.fr_from_int(srs_trapdoor))
            g = zo.fixed_base_mul(mono, threads)
            com = lambda col: zo.affine_to_ints(zo.g1_to_affine(zo.best_multiexp(np.asarray(b.to_host(col), dtype=np.uint64), g, threads)).reshape(1, 8))[0]
            fixed, sigma = [com(c) for c in prover.fixed_coeff], [com(c) for c in prover.sigma_coeff]
        else:
            fixed = [_pts([c[0]])[0] for c in b.commit(prover.fixed_coeff, lagrange=False)]
            sigma = [_pts([c[0]])[0] for c in b.commit(prover.sigma_coeff, lagrange=False)]
        setattr(prover, key, (fixed, sigma))
    return getattr(prover, key)


def srs_g2_from_params_file(path):
    """(g2, s_g2) of a ParamsKZG file: its last 256 bytes (poly/kzg/commitment.rs ParamsKZG::write order: k, g, g_lagrange, g2, s_g2)"""
    raw = open(path, "rb").read()[-256:]
    return P.g2_from_raw_bytes(raw[:128]), P.g2_from_raw_bytes(raw[128:])


def verify_proof(prover, wit, proof, kind, srs_trapdoor=0x1D5C0FFEE, oracle_vk=False, srs_g2=None):
    """True iff the proof BYTES verify: pyref.verify_proof_bytes reads them in upstream's verifier order with the named transcript
    ("blake2b", "evm", "poseidon"), re-derives every challenge, and checks the gate / permutation / lookup identities and the
    SHPLONK opening.  Nothing of the prover's trace is consulted.  srs_g2 = (g2, s_g2) from the params file: the opening is closed by the
    PAIRING e(L, g2) = e(h2, s_g2) — the reference's own acceptance check (evm_verify, /root/reference/src/bin/cli.rs:524) — and the
    trapdoor is not used at all; else by the same equation in G1 under the known trapdoor of a synthetic SRS."""
    import halo2_zkcert_amd.prover as pv

    sh = prover.shape
    aq, fq = _queries(sh)
    fixed, sigma = vk_commitments(prover, oracle_side=oracle_vk, srs_trapdoor=srs_trapdoor)
    inst_vals = [zo.fr_arr_to_ints(np.asarray(v, dtype=np.uint64)) for v in wit["instance_values"]]
    inst_cols = [v + [0] * ((1 << sh.k) - len(v)) for v in inst_vals]
    return P.verify_proof_bytes(_vk(sh), kind, proof, pv.from_mont_host(prover.vk_repr), inst_vals, inst_cols, fixed, sigma, aq, fq,
                                None if srs_g2 is not None else srs_trapdoor, srs_g2)


def verify_trace(prover, wit, trace, srs_trapdoor=0x1D5C0FFEE, tamper=None):
    """True iff the proof in `trace` verifies.  The vk part (fixed / sigma commitments) is committed here with the prover's
    own backend, as keygen would.  tamper(evals, commitments) may corrupt the proof first."""
    sh, b = prover.shape, prover.b
    vk = _vk(sh)
    pts = trace["points"]
    coms = {}
    for i, p_ in zip(sh.advice_commit_order(), _pts(pts["advice"])):      # transcript order: by phase, by column inside a phase
        coms[("advice", i)] = p_
    L = len(sh.lookups)
    lp = _pts(pts.get("lookup_permuted", []))
    for i in range(L):      # transcript order: (permuted input, permuted table) per lookup
        coms[("lookup_a", i)], coms[("lookup_s", i)] = lp[2 * i], lp[2 * i + 1]
    prods = _pts(pts["products"])
    for i in range(sh.n_perm_sets):
        coms[("perm_z", i)] = prods[i]
    for i in range(L):
        coms[("lookup_z", i)] = prods[sh.n_perm_sets + i]
    coms[("random", 0)] = _pts(pts["random_poly"])[0]
    fixed, sigma = vk_commitments(prover)
    for i, c in enumerate(fixed):
        coms[("fixed", i)] = c
    for i, c in enumerate(sigma):
        coms[("sigma", i)] = c
    h_pieces = _pts(pts["quotient"])
    import halo2_zkcert_amd.prover as pv

    evals = {q: v for q, v in pv.eval_ints(trace).items() if q[0] != ("h", 0)}
    instance = [zo.fr_arr_to_ints(b.to_host(c)) for c in wit["instance"]]
    h1, h2 = _pts(pts["shplonk_h1"])[0], _pts(pts["shplonk_h2"])[0]
    kind = trace.get("transcript", "blake2b-py")
    if kind != "evm":
        # the proof's bytes are the compressed forms of the points the transcript absorbed
        seen = {}
        for tag, hx in trace["commitments"]:
            i = seen.get(tag, 0)
            seen[tag] = i + 1
            assert zo.g1_to_bytes(np.asarray(pts[tag][i], dtype=np.uint64)).hex() == hx, (tag, i)
    # the verifier's side of Fiat-Shamir: replay the transcript over the proof and require the prover's challenges
    ts = pv.make_transcript(kind)
    ch = trace["challenges"]
    ts.common_scalar(prover.vk_repr)
    for col in wit["instance_values"]:
        for v in col:
            ts.common_scalar(v)
    order, user, at = sh.advice_commit_order(), [], 0
    for ph in sh.phases:
        for _ in [i for i in order if sh.advice_phase[i] == ph]:
            ts.write_point(pts["advice"][at])
            at += 1
        user += [ts.squeeze() for p_ in sh.challenge_phase if p_ == ph]
    assert user == list(ch.get("user", [])), (user, ch.get("user"))
    assert ts.squeeze() == ch["theta"]
    for a in pts.get("lookup_permuted", []):
        ts.write_point(a)
    assert ts.squeeze() == ch["beta"] and ts.squeeze() == ch["gamma"]
    for a in pts["products"] + pts["random_poly"]:
        ts.write_point(a)
    assert ts.squeeze() == ch["y"]
    for a in pts["quotient"]:
        ts.write_point(a)
    assert ts.squeeze() == ch["x"]
    rows = {q: row for q, row in trace["evals"]}
    for q in trace["eval_write_order"]:
        ts.write_scalar(rows[q])
    assert ts.squeeze() == ch["shplonk_y"] and ts.squeeze() == ch["shplonk_v"]
    ts.write_point(pts["shplonk_h1"][0])
    assert ts.squeeze() == ch["shplonk_u"]
    if tamper:
        tamper(evals, coms, instance)
    return P.plonk_verify(vk, instance, coms, h_pieces, evals, trace["query_list"], trace["challenges"], h1, h2, srs_trapdoor)
