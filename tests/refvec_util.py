"""The `prover` section of tests/golden/reference_vectors.json (integration/rust/refvec/src/prover_vectors.rs, run against the UNPATCHED
crates the reference pins: /root/reference/Cargo.lock:1320-1322,1359-1361,2714-2716) — consumer side (TEST INFRASTRUCTURE).

  * build_circuit(two_phase): the explicit circuit instance of prover_vectors.rs `ShapeCircuit::build`, restated (the self-made schema file
    of tests/test_refvec_schema.py is written from it; the real file carries upstream's own copy and nothing here is regenerated then);
  * RngStream / draw_roles: the SplitMix64 stream `CountingRng` hands to create_proof and upstream's draw order [UPSTREAM-RECALL], in the
    two candidate forms this repository knows — "kzg" (only the values KZG uses: blinding rows, the random polynomial) and "blinds" (the same
    plus one `Blind(Scalar::random)` per commitment, which KZG draws and ignores in the un-forked prover).  The u64 count printed by the
    Rust side decides which one the pinned fork follows; a count that fits neither fails with both counts in the message;
  * prover_and_witness(backend, doc, roles): a prover.Prover over explicit key data (Prover.from_explicit) and its witness;
  * emit_prover_section(...): the same schema written from THIS repository's oracle (labelled self-made; pins nothing).
"""
import hashlib

import numpy as np

R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
K = 6
SRS_TRAPDOOR = 0x1D5C0FFEE
RNG_SEED = 0x5EED0006
M64 = (1 << 64) - 1
H = lambda s: int(s, 16)
hx = lambda x: format(int(x), "064x")


def _splitmix_next(state):
    state = (state + 0x9E3779B97F4A7C15) & M64
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return state, z ^ (z >> 31)


class RngStream:
    """prover_vectors.rs CountingRng: the SplitMix64 sequence; fr() = one Fr::random = 8 u64, little-endian 512-bit value mod r"""

    def __init__(self, seed=RNG_SEED):
        self.state, self.u64_drawn = seed, 0

    def u64(self):
        self.state, out = _splitmix_next(self.state)
        self.u64_drawn += 1
        return out

    def fr(self):
        v = 0
        for i in range(8):
            v |= self.u64() << (64 * i)
        return v % R

    def frs(self, count):
        return [self.fr() for _ in range(count)]


def build_circuit(two_phase):
    """prover_vectors.rs ShapeCircuit::build, value for value -> the dict its describe() prints (hex strings)"""
    n, usable = 1 << K, (1 << K) - 7
    st = [0xC1C017 + (1 if two_phase else 0)]

    def small(bound):
        st[0], out = _splitmix_next(st[0])
        return out % bound
    gates = list(range(0, usable - 3, 4))
    fixed = [[0] * usable for _ in range(4)]
    for r in gates:
        fixed[0][r] = fixed[1][r] = 1
    for i in range(8):
        fixed[2][i] = 100 + i
    for i in range(usable):
        fixed[3][i] = i
    advice = [[0] * usable for _ in range(3)]
    for i in range(usable):
        advice[0][i] = small(1 << 40)
        advice[1][i] = small(1 << 40)
        advice[2][i] = small(usable)
    instance = [1000 + j for j in range(4)]
    perm_instance = 5 if two_phase else 4
    perm_constants = perm_instance - 1
    copies = []
    for g in range(4):
        advice[0][gates[g] + 1] = advice[2][g]
        copies.append([0, gates[g] + 1, 2, g])
        advice[1][gates[g] + 1] = fixed[2][g]
        copies.append([1, gates[g] + 1, perm_constants, g])
    for j, g in enumerate(range(4, 8)):
        advice[0][gates[g] + 2] = instance[j]
        copies.append([0, gates[g] + 2, perm_instance, j])
    for r in gates:
        advice[0][r + 3] = (advice[0][r] + advice[0][r + 1] * advice[0][r + 2]) % R
    for g in range(8, 11):
        advice[1][gates[g] + 2] = advice[0][gates[g] + 3]
        copies.append([1, gates[g] + 2, 0, gates[g] + 3])
    for r in gates:
        advice[1][r + 3] = (advice[1][r] + advice[1][r + 1] * advice[1][r + 2]) % R
    col = lambda v: [hx(x) for x in v]
    return {"k": K, "two_phase": two_phase, "usable_rows": usable, "fixed": [col(c) for c in fixed], "advice": [col(c) for c in advice],
            "instance": [col(instance)], "copies": copies}


def shape_of(pv, circuit):
    sh = pv.CircuitShape.two_phase(circuit["k"]) if circuit["two_phase"] else pv.CircuitShape.small(circuit["k"])
    assert (1 << sh.k) - (sh.blinding_factors + 1) == circuit["usable_rows"]
    return sh


def expected_fr_draws(sh, model):
    """how many Fr::random calls one create_proof makes under each draw-order model"""
    bf, L, Zp, n = sh.blinding_factors, len(sh.lookups), sh.n_perm_sets, 1 << sh.k
    values = sh.n_advice * (bf + 1) + L * 2 * (bf + 1) + Zp * bf + L * bf + n
    if model == "kzg":
        return values
    if model == "blinds":      # + one Blind per advice / permuted input / permuted table / z / random-polynomial / quotient-piece commitment
        return values + sh.n_advice + 2 * L + Zp + L + 1 + (sh.degree - 1)
    raise ValueError(model)


def draw_roles(sh, model, seed=RNG_SEED):
    """Replays create_proof's draws in upstream's order [UPSTREAM-RECALL: plonk/prover.rs — per phase: the blinding rows of that phase's advice
    columns (then, model "blinds", one Blind per column); lookup by lookup: the permuted input's and the permuted table's blinding rows
    (then two Blinds); set by set: the permutation z's blinding rows (then a Blind); lookup by lookup: z's blinding rows (then a Blind); the
    random polynomial's n coefficients (then a Blind); (then one Blind per quotient piece)] -> dict of canonical ints per role + u64_drawn"""
    rng = RngStream(seed)
    bf, L, Zp, n = sh.blinding_factors, len(sh.lookups), sh.n_perm_sets, 1 << sh.k
    blind = (lambda count: rng.frs(count)) if model == "blinds" else (lambda count: [])
    advice = {}
    for ph in sh.phases:
        cols = [c for c in range(sh.n_advice) if sh.advice_phase[c] == ph]
        for c in cols:
            advice[c] = rng.frs(bf + 1)
        blind(len(cols))
    lookup_permuted = []
    for _ in range(L):
        lookup_permuted += rng.frs(bf + 1)      # permuted input
        lookup_permuted += rng.frs(bf + 1)      # permuted table
        blind(2)
    perm_z = []
    for _ in range(Zp):
        perm_z += rng.frs(bf)
        blind(1)
    lookup_z = []
    for _ in range(L):
        lookup_z += rng.frs(bf)
        blind(1)
    random_poly = rng.frs(n)
    blind(1)
    blind(sh.degree - 1)
    return dict(advice=advice, lookup_permuted=lookup_permuted, perm_z=perm_z, lookup_z=lookup_z, random_poly=random_poly, u64_drawn=rng.u64_drawn)


def pick_model(sh, u64_drawn):
    for model in ("kzg", "blinds"):
        if 8 * expected_fr_draws(sh, model) == u64_drawn:
            return model
    raise AssertionError(f"create_proof drew {u64_drawn} u64 from its rng; the draw orders this repository knows take "
                         f"{8 * expected_fr_draws(sh, 'kzg')} (KZG values only) or {8 * expected_fr_draws(sh, 'blinds')} (plus one Blind per commitment): "
                         "upstream's order is another one — restate draw_roles() from plonk/prover.rs of the pinned revision")


def prover_and_witness(pv, zo, backend, doc, roles):
    """doc: one of the file's `small` / `two_phase` objects -> (Prover over the file's explicit key data, witness, blinding dict)"""
    c = doc["circuit"]
    sh = shape_of(pv, c)
    n, u = 1 << sh.k, c["usable_rows"]
    col = lambda hexes: zo.fr_arr_from_ints([H(x) for x in hexes] + [0] * (n - len(hexes)))
    vk_repr = zo.fr_from_int(H(doc["vk_transcript_repr"])) if doc.get("vk_transcript_repr") else None
    p = pv.Prover.from_explicit(backend, sh, [col(f) for f in c["fixed"]], [tuple(x) for x in c["copies"]], srs_trapdoor=H(doc.get("srs_trapdoor", hx(SRS_TRAPDOOR))),
                                vk_repr=vk_repr)
    b = backend
    advice = []
    for j in range(sh.n_advice):
        if sh.advice_phase[j] == 0:
            vals = [H(x) for x in c["advice"][j]] + roles["advice"][j]
        else:
            vals = [0] * n                   # synthesised once the phase's challenges exist (advice_for_phase below)
        assert len(vals) == n
        advice.append(b.from_host(zo.fr_arr_from_ints(vals)))
    inst_vals = [zo.fr_arr_from_ints([H(x) for x in col_]) for col_ in c["instance"]]
    instance = []
    for v in inst_vals:
        full = np.zeros((n, 4), dtype=np.uint64)
        full[:len(v)] = v
        instance.append(b.from_host(full))

    def advice_for_phase(wit, phase, user_challenges):      # prover_vectors.rs: a3 = challenge_0 * a0 on the usable rows; blinding rows from the rng
        for j in range(sh.n_advice):
            if sh.advice_phase[j] == phase and phase > 0:
                a0 = [H(x) for x in c["advice"][0]]
                vals = [a * user_challenges[0] % R for a in a0] + roles["advice"][j]
                wit["advice"][j] = b.from_host(zo.fr_arr_from_ints(vals))
    wit = dict(advice=advice, instance=instance, instance_values=inst_vals, base=0, advice_for_phase=advice_for_phase if c["two_phase"] else None)
    fr = zo.fr_arr_from_ints
    blinding = dict(lookup_permuted=fr(roles["lookup_permuted"]), perm_z=fr(roles["perm_z"]), lookup_z=fr(roles["lookup_z"]), random_poly=fr(roles["random_poly"]))
    return p, wit, blinding


def check_constraint_system(pv, doc):
    """upstream's own description of the constraint system against the shape the repository proves (column counts, degree, blinding factors,
    the query lists in order, the permutation columns in order)"""
    sh, cs = shape_of(pv, doc["circuit"]), doc["constraint_system"]
    assert cs["degree"] == sh.degree and cs["blinding_factors"] == sh.blinding_factors, (cs["degree"], cs["blinding_factors"])
    assert (cs["num_fixed_columns"], cs["num_advice_columns"], cs["num_instance_columns"]) == (sh.n_fixed, sh.n_advice, sh.n_instance)
    assert cs["num_selectors"] == 0
    assert [tuple(q) for q in cs["advice_queries"]] == [(c, r) for kind, c, r in sh.queries() if kind == "advice"]
    assert [tuple(q) for q in cs["fixed_queries"]] == [(c, r) for kind, c, r in sh.queries() if kind == "fixed"]
    assert [tuple(q) for q in cs["instance_queries"]] == [(c, r) for kind, c, r in sh.queries() if kind == "instance"]
    kinds = [str(t).lower() for t, _ in cs["permutation_columns"]]
    assert [i for _, i in cs["permutation_columns"]] == [i for _, i in sh.perm_columns]
    assert all(k_.startswith(want) for k_, (want, _) in zip(kinds, sh.perm_columns)), kinds


# ------------------------------------------------------------------------------------------------------------------------ the self-made file
def emit_prover_section(pv, zo, ffi, fm, verify_proof, vk_commitments, tmp_dir, threads=2):
    """the `prover` object of reference_vectors.json written from this repository's ORACLE (schema check; pins nothing)"""
    from oracle_backend import OracleBackend

    def one(two_phase):
        circuit = build_circuit(two_phase)
        sh = shape_of(pv, circuit)
        roles = draw_roles(sh, "kzg")
        doc = {"circuit": circuit, "vk_transcript_repr": hx(int.from_bytes(hashlib.blake2b(sh.name.encode(), digest_size=64).digest(), "little") % R)}
        p, wit, blinding = prover_and_witness(pv, zo, OracleBackend(threads), doc, roles)
        fixed, sigma = vk_commitments(p)
        pt = lambda xy: zo.g1_to_bytes(zo.affine_from_ints([xy])[0]).hex()
        doc["constraint_system"] = {
            "degree": sh.degree, "blinding_factors": sh.blinding_factors, "num_fixed_columns": sh.n_fixed, "num_advice_columns": sh.n_advice,
            "num_instance_columns": sh.n_instance, "num_selectors": 0,
            "advice_queries": [[c, r] for kind, c, r in sh.queries() if kind == "advice"],
            "fixed_queries": [[c, r] for kind, c, r in sh.queries() if kind == "fixed"],
            "instance_queries": [[c, r] for kind, c, r in sh.queries() if kind == "instance"],
            "permutation_columns": [[t.capitalize(), i] for t, i in sh.perm_columns], "permutation_columns_note": "self-made"}
        doc["fixed_commitments"], doc["permutation_commitments"] = [pt(x) for x in fixed], [pt(x) for x in sigma]
        doc["proofs"] = {}
        for kind in ("blake2b", "poseidon", "evm"):
            w = dict(wit, advice=list(wit["advice"]))
            t = p.prove(w, transcript=kind, blinding=blinding)
            order = [("user", i) for i in range(len(sh.challenge_phase))] + ["theta", "beta", "gamma", "y", "x", "shplonk_y", "shplonk_v", "shplonk_u"]
            chs = [t["challenges"]["user"][tag[1]] if isinstance(tag, tuple) else t["challenges"][tag] for tag in order]
            doc["proofs"][kind] = {"proof": bytes(t["proof"]).hex(), "challenges": [hx(c) for c in chs], "rng_u64_drawn": roles["u64_drawn"],
                                   "verified": bool(verify_proof(p, w, t["proof"], kind))}
        return doc, p, wit, blinding
    small, p, wit, blinding = one(False)
    two_phase = one(True)[0]
    # files: ParamsKZG::write, ProvingKey::write(RawBytesUnchecked), bincode(Snark) — this repository's writers
    sh = p.shape
    g2 = ffi._g2_setup_bytes(SRS_TRAPDOOR)
    params = fm.ParamsFile(K, np.asarray(p.b.g, dtype=np.uint64).reshape(-1, 8), np.asarray(p.b.g_lagrange, dtype=np.uint64).reshape(-1, 8), g2).to_bytes()
    fixed, sigma = vk_commitments(p)
    aff = lambda pts: np.stack([zo.affine_from_ints([xy])[0] for xy in pts])
    pk_path = tmp_dir / "self_made.pk"
    fm.ProvingKeyFile.from_prover(p, aff(fixed), aff(sigma)).write(pk_path)
    proof = bytes(p.prove(dict(wit, advice=list(wit["advice"])), transcript="poseidon", blinding=blinding)["proof"])
    inst = [[H(x) for x in c] for c in small["circuit"]["instance"]]
    snark = fm.SnarkFile(bytes((i * 37 + 11) % 251 + 1 for i in range(96)), inst, proof)      # an opaque stand-in for the bincode protocol
    files = {"params": params.hex(), "pk": open(pk_path, "rb").read().hex(), "pk_n_selectors": 0, "snark": (snark.protocol + snark.tail_bytes()).hex(),
             "snark_protocol_len": len(snark.protocol), "snark_proof": proof.hex(), "snark_instances": [[hx(v) for v in c] for c in inst], "snark_note": "self-made"}
    return {"srs_trapdoor": format(SRS_TRAPDOOR, "x"), "rng_seed": RNG_SEED, "rng_spec": "self-made", "evaluate_h_note": "self-made", "small": small,
            "two_phase": two_phase, "files": files}
