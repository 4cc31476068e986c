"""Runs oracle/pyref.py's algebraic PLONK + SHPLONK verifier over a Prover trace (TEST INFRASTRUCTURE)."""
import numpy as np

import pyref as P
import zkoracle_py as zo


def _pts(arrs):
    return [zo.affine_to_ints(np.asarray(a, dtype=np.uint64).reshape(1, 8))[0] for a in arrs]


def verify_trace(prover, wit, trace, srs_trapdoor=0x1D5C0FFEE, tamper=None):
    """True iff the proof in `trace` verifies.  The vk part (fixed / sigma commitments) is committed here with the prover's
    own backend, as keygen would.  tamper(evals, commitments) may corrupt the proof first."""
    sh, b = prover.shape, prover.b
    vk = dict(k=sh.k, degree=sh.degree, blinding_factors=sh.blinding_factors, gates=sh.gates, lookups=sh.lookups,
              perm_columns=sh.perm_columns)
    pts = trace["points"]
    coms = {}
    for i, p_ in enumerate(_pts(pts["advice"])):
        coms[("advice", i)] = p_
    L = len(sh.lookups)
    lp = _pts(pts.get("lookup_permuted", []))
    for i in range(L):
        coms[("lookup_a", i)], coms[("lookup_s", i)] = lp[i], lp[L + i]
    prods = _pts(pts["products"])
    for i in range(sh.n_perm_sets):
        coms[("perm_z", i)] = prods[i]
    for i in range(L):
        coms[("lookup_z", i)] = prods[sh.n_perm_sets + i]
    coms[("random", 0)] = _pts(pts["random_poly"])[0]
    for i, c in enumerate(b.commit(prover.fixed_coeff, lagrange=False)):
        coms[("fixed", i)] = _pts([c[0]])[0]
    for i, c in enumerate(b.commit(prover.sigma_coeff, lagrange=False)):
        coms[("sigma", i)] = _pts([c[0]])[0]
    h_pieces = _pts(pts["quotient"])
    import halo2_zkcert_amd.prover as pv

    evals = {q: v for q, v in pv.eval_ints(trace).items() if q[0] != ("h", 0)}
    instance = [zo.fr_arr_to_ints(b.to_host(c)) for c in wit["instance"]]
    h1, h2 = _pts(pts["shplonk_h1"])[0], _pts(pts["shplonk_h2"])[0]
    # the proof's bytes are the compressed forms of the points the transcript absorbed
    seen = {}
    for tag, hx in trace["commitments"]:
        i = seen.get(tag, 0)
        seen[tag] = i + 1
        assert zo.g1_to_bytes(np.asarray(pts[tag][i], dtype=np.uint64)).hex() == hx, (tag, i)
    # the verifier's side of Fiat-Shamir: replay the transcript over the proof and require the prover's challenges
    import halo2_zkcert_amd.prover as pv2

    ts = pv2.Blake2bTranscript()
    ch = trace["challenges"]
    for a in pts["advice"]:
        ts.write_point(a)
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
    for q, row in trace["evals"]:
        if q[0] != ("h", 0):
            ts.write_scalar(row)
    assert ts.squeeze() == ch["shplonk_y"] and ts.squeeze() == ch["shplonk_v"]
    ts.write_point(pts["shplonk_h1"][0])
    assert ts.squeeze() == ch["shplonk_u"]
    if tamper:
        tamper(evals, coms, instance)
    return P.plonk_verify(vk, instance, coms, h_pieces, evals, trace["query_list"], trace["challenges"], h1, h2, srs_trapdoor)
