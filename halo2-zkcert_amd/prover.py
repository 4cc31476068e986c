"""The create_proof-shaped schedule over the hot-path kernels, on synthetic 2^k-row tables.

Mirrors the order of operations of halo2_proofs::plonk::create_proof
[UPSTREAM-RECALL src/plonk/prover.rs; crate pinned at /root/reference/Cargo.lock:1320-1322; reached
from gen_snark_shplonk at /root/reference/src/helpers.rs:233,299 and src/bin/cli.rs:320,343,369,462]:

  1 commit advice (A MSMs over g_lagrange)                          -> transcript -> theta
  2 lookup permuted columns: 2L iNTT_n + 2L MSM_n                   -> transcript -> beta, gamma
  3 permutation / lookup grand products: (Zp + L) iNTT_n + MSM_n    -> transcript
  4 vanishing random poly: 1 MSM_n                                  -> transcript -> y
  5 advice/instance iNTT_n; (A+I+3L+Zp) coset NTT_{e n}; quotient sweep; / (X^n-1); iNTT_{e n};
    q MSM_n over g                                                  -> transcript -> x
  6 SHPLONK multi-open of every queried polynomial: linear combinations, kate divisions, 2 MSM_n

Also computed for real: the theta-compression of the lookup expressions, the permutation and
lookup grand products (batch inversion + running product; the blinding rows are seeded stand-ins for the rng),
and the evaluations of every queried polynomial at x * omega^rotation.

The lookup argument's permuted columns are computed for real too (permute_expression_pair: a sort), which is why
the synthetic lookup-advice columns draw their values from the table column: the lookup has to be satisfiable.

Transcript operations follow upstream's order (zkhip.h, zkhip_create_proof_ex): vk.transcript_repr and the instance values are
absorbed first; the permuted lookup commitments go in per lookup (input, table); the evaluations are WRITTEN in upstream's order
(advice, fixed, random, sigma, per permutation set, per lookup) while the multi-open consumes them in upstream's QUERY order.
Transcripts: Blake2bTranscript below (halo2's Blake2bWrite over hashlib) or the library's Blake2b / Poseidon / Keccak transcripts
driven from here (make_transcript): Poseidon is what the reference's gen_snark_shplonk commands use, Keccak what
gen_evm_proof_shplonk uses.

What is NOT here (out of scope): witness synthesis — the witness is a synthetic satisfiable instance of the circuit's shape
(Prover._build_satisfiable / _build_satisfiable_sha), so the outputs are valid proofs.

The schedule is written against a small backend interface so the same code drives the HIP library
(GpuBackend, here) and, in tests/ and bench.py's cpu_baseline leg only, the CPU oracle.
"""
import contextlib
import hashlib
import os
import sys

import time

import numpy as np

from . import evaluator as ev
from .shplonk import ProverSHPLONK as ShplonkProver

R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
R_INV_256 = pow(1 << 256, -1, R)


class CircuitShape:
    """Column / gate shape of one of the reference's circuits."""

    def __init__(self, name, k, n_basic_advice, n_lookup_advice, n_instance, degree, blinding_factors, seed, gates=None,
                 n_fixed=None, perm_columns=None, n_phase1=0):
        self.name, self.k, self.seed = name, k, seed
        self.layout = "halo2-lib"
        self.n_basic, self.n_lookup = n_basic_advice, n_lookup_advice
        # Advice PHASES (axiom's halo2 create_proof [UPSTREAM-RECALL]: the advice columns are committed phase by phase and the user
        # challenges of a phase are squeezed after its commitments; halo2-lib's RLC chips put their running sums in phase 1).  The
        # reference's three circuits are phase 0 only; n_phase1 appends columns of phase 1, each a(x) = challenge_0 * advice_0(x)
        # (gate: selector_0 * (a - advice_0 * challenge_0)), with ONE user challenge squeezed after the phase-0 commitments.
        self.n_phase1 = n_phase1
        self.n_advice = n_basic_advice + n_lookup_advice + n_phase1
        self.advice_phase = [0] * (n_basic_advice + n_lookup_advice) + [1] * n_phase1
        self.challenge_phase = [0] if n_phase1 else []
        self.n_instance = n_instance
        # fixed: one selector per basic advice column, one constants column, one lookup table column
        self.n_fixed = n_basic_advice + 1 + (1 if n_lookup_advice else 0)
        self.degree, self.blinding_factors = degree, blinding_factors
        A = lambda c, r: ("advice", c, r)
        # halo2-lib FlexGateConfig vertical gate on every basic advice column: q * (a + b*c - d)
        self.gates = [("prod", ("fixed", c, 0), ("sum", ("sum", A(c, 0), ("prod", A(c, 1), A(c, 2))), ("neg", A(c, 3))))
                      for c in range(n_basic_advice)]
        # halo2-lib RangeConfig: lookup_advice in the [0, 2^lookup_bits) table column
        self.lookups = [([A(n_basic_advice + i, 0)], [("fixed", self.n_fixed - 1, 0)]) for i in range(n_lookup_advice)]
        self.perm_columns = ([("advice", c) for c in range(self.n_advice)] + [("fixed", n_basic_advice)]
                             + [("instance", i) for i in range(n_instance)])
        for j in range(n_phase1):
            c1 = n_basic_advice + n_lookup_advice + j
            self.gates.append(("prod", ("fixed", 0, 0), ("sum", A(c1, 0), ("neg", ("prod", A(0, 0), ("challenge", 0))))))
        if gates is not None:
            self.gates = gates
        if n_fixed is not None:
            self.n_fixed = n_fixed
        if perm_columns is not None:
            self.perm_columns = perm_columns
        chunk = degree - 2
        self.n_perm_sets = -(-len(self.perm_columns) // chunk)

    @classmethod
    def rsa(cls, k=17):
        """RSA circuit, README row k=17: 3 advice + 1 lookup-advice + 1 fixed(constants) (+ selectors, table)."""
        basic = {15: 12, 16: 6, 17: 3}.get(k, 3)
        return cls(f"rsa_k{k}", k, basic, 1, 1, 4, 6, 0xC0FFEE00 + k)

    @classmethod
    def agg(cls, k=22, n_basic=3, n_lookup=1):
        """X509VerifierAggregationCircuit's shape (BASELINE configs[3]; /root/reference/src/bin/cli.rs:464-527): a halo2-lib
        BaseCircuitBuilder circuit at k = 22 with lookup_bits = k - 1 (cli.rs:475) — vertical-gate advice columns, lookup-advice
        columns, one constants column, one instance column.  The column counts are what calculate_params(Some(10)) returns for four
        verified snarks (cli.rs:493), which the reference does not state: parameters here."""
        return cls(f"agg_k{k}_a{n_basic}+{n_lookup}", k, n_basic, n_lookup, 1, 4, 6, 0xA6600000 + k)

    @classmethod
    def sha256(cls, k=19, n_advice=32, n_fixed=12):
        """zkEVM SHA-256 bit circuit shape: many narrow bit/word columns, fixed q_* and
        round-constant columns, boolean and word-decomposition gates of degree up to 5, no lookup; the two digest words are
        the only permutation (instance copy) columns.  Column counts are parameters (the real ones are in zkevm-hashes)."""
        A = lambda c, r: ("advice", c, r)
        F = lambda c, r=0: ("fixed", c, r)
        one = ("const", 1)
        gates = []
        for c in range(n_advice):
            q = F(c % n_fixed)
            if c % 4 == 0:      # q * b * (1 - b): bit columns
                gates.append(("prod", q, ("prod", A(c, 0), ("sum", one, ("neg", A(c, 0))))))
            elif c % 4 == 1:    # q * (a ^ b ^ c as a degree-3 polynomial over bits of neighbouring rows)
                x, y, z = A(c, 0), A(c - 1, 0), A(c - 1, 1)
                xy = ("prod", x, y)
                gates.append(("prod", q, ("sum", A(c, 1), ("neg", ("sum", ("sum", x, y), ("scaled", ("prod", xy, z), 4))))))
            elif c % 4 == 2:    # word recomposition over rotations: q * (w - sum 2^i b_i)
                acc = A(c - 2, -3)
                for i, r in enumerate((-2, -1, 0, 1, 2, 3)):
                    acc = ("sum", ("scaled", acc, 2), A(c - 2, r))
                gates.append(("prod", q, ("sum", A(c, 0), ("neg", acc))))
            else:               # round-constant addition with a carry bit, degree 5 with two selectors
                t = ("sum", ("sum", A(c, 0), A(c - 1, 0)), F((c + 1) % n_fixed))
                gates.append(("prod", ("prod", q, F((c + 5) % n_fixed)), ("prod", t, ("prod", A(c - 3, 0), ("sum", one, ("neg", A(c - 3, 1)))))))
        sh = cls(f"sha256_k{k}", k, n_advice, 0, 1, 5, 6, 0x5A256000 + k, gates=gates, n_fixed=n_fixed,
                 perm_columns=[("advice", 0), ("advice", 1), ("instance", 0)])
        sh.layout = "sha"
        return sh

    @classmethod
    def small(cls, k=8):
        return cls(f"small_k{k}", k, 2, 1, 1, 4, 6, 0x5EED00 + k)

    @classmethod
    def two_phase(cls, k=8, n_phase1=1):
        """the small circuit plus advice columns of the SECOND phase and one user challenge (see __init__)"""
        return cls(f"two_phase_k{k}_p{n_phase1}", k, 2, 1, 1, 4, 6, 0x2F4A5E00 + k, n_phase1=n_phase1)

    @property
    def phases(self):
        return sorted(set(self.advice_phase) | set(self.challenge_phase)) or [0]

    def advice_commit_order(self):
        """advice column indices in the order their commitments enter the transcript: by phase, by column inside a phase"""
        return [i for ph in self.phases for i in range(self.n_advice) if self.advice_phase[i] == ph]

    def queries(self):
        """Distinct (kind, column, rotation) queries of the gates and lookups (what create_proof evaluates at x)."""
        if getattr(self, "_queries", None) is not None:
            return self._queries
        seen, out = set(), []

        def walk(e):
            if e[0] in ("advice", "fixed", "instance"):
                if e not in seen:
                    seen.add(e)
                    out.append(e)
            else:
                for c in e[1:]:
                    if isinstance(c, tuple):
                        walk(c)

        for g in self.gates:
            walk(g)
        for ins, tabs in self.lookups:
            for e in ins + tabs:
                walk(e)
        for t, i in self.perm_columns:
            walk((t, i, 0))
        self._queries = out
        return out

    def counts(self, dom_extended_k):
        L, Zp, A, I = len(self.lookups), self.n_perm_sets, self.n_advice, self.n_instance
        q = self.degree - 1
        return dict(msm=A + 3 * L + Zp + 1 + q + 2, intt_n=A + I + 3 * L + Zp, ntt_ext=A + I + 3 * L + Zp, intt_ext=1,
                    sweep_rows=1 << dom_extended_k)


Q_MOD = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47   # Fq
Q_INV_256 = pow(1 << 256, -1, Q_MOD)


class Blake2bTranscript:
    """halo2_proofs transcript.rs Blake2bWrite<_, G1Affine, Challenge255<_>> [UPSTREAM-RECALL; crate pinned at
    /root/reference/Cargo.lock:1320-1322]: a BLAKE2b-512 state personalised "Halo2-Transcript"; a point is absorbed as prefix 1
    and its canonical x and y (32 little-endian bytes each), a scalar as prefix 2 and its canonical 32 bytes; a challenge is
    prefix 0 followed by the 64-byte digest of a clone of the state, reduced into Fr as a little-endian integer.
    (The reference's own commands use snark-verifier's Poseidon / Keccak transcripts: those live in the library, see
    make_transcript; this class is the hashlib cross-check of the library's Blake2b one.)"""

    def __init__(self):
        self.state = hashlib.blake2b(digest_size=64, person=b"Halo2-Transcript")

    @staticmethod
    def _canon(limbs, inv, mod):
        v = int(limbs[0]) | int(limbs[1]) << 64 | int(limbs[2]) << 128 | int(limbs[3]) << 192
        return (v * inv % mod).to_bytes(32, "little")

    def write_point(self, xy):
        """xy: affine point, 8 Montgomery limbs (x, y); the identity is (0, 0)"""
        self.state.update(b"\x01" + self._canon(xy[:4], Q_INV_256, Q_MOD) + self._canon(xy[4:], Q_INV_256, Q_MOD))

    def write_scalar(self, limbs):
        self.state.update(b"\x02" + self._canon(limbs, R_INV_256, R))

    common_scalar = write_scalar      # absorbed the same way; only the proof stream differs (and this class keeps none)

    def squeeze(self):
        self.state.update(b"\x00")
        return int.from_bytes(self.state.copy().digest(), "little") % R


class LibTranscriptAdapter:
    """The library's ready-made transcripts (ffi.LibTranscript: "blake2b", "evm", "poseidon") behind the interface the Python schedule
    uses (write_point / write_scalar / common_scalar on ABI limbs, squeeze() -> canonical int)."""

    def __init__(self, kind):
        from . import ffi

        self.t = ffi.LibTranscript(kind)

    def write_point(self, xy):
        self.t.write_point(np.ascontiguousarray(xy, dtype=np.uint64))

    def write_scalar(self, limbs):
        self.t.write_scalar(limbs)

    def common_scalar(self, limbs):
        self.t.common_scalar(limbs)

    def squeeze(self):
        return from_mont_host(self.t.squeeze_limbs())

    def proof(self):
        return self.t.proof()


def make_transcript(kind):
    """"blake2b-py": Blake2bTranscript above (hashlib); "blake2b" / "evm" / "poseidon": the library's"""
    return Blake2bTranscript() if kind == "blake2b-py" else LibTranscriptAdapter(kind)


def challenge(tag, commitment_bytes):
    """Stand-in Fiat-Shamir squeeze: BLAKE2b(tag || commitments) reduced into Fr (canonical int)."""
    h = hashlib.blake2b(tag.encode() + b"".join(commitment_bytes), digest_size=64).digest()
    return int.from_bytes(h, "little") % R


class GpuBackend:
    """Drives libzkhip.so; columns are (n, 4) int64 torch tensors resident in HBM."""

    def __init__(self, ctx, ffi):
        self.ctx, self.ffi = ctx, ffi
        self.torch = ctx.torch
        # the proof's dependent chain runs on a high-priority stream, the coset NTTs that overlap it on a normal one: the
        # latency-bound MSM phases then get their CUs first and the NTTs fill what is idle
        if self.torch.cuda.current_stream(ctx.device) == self.torch.cuda.default_stream(ctx.device):
            self.torch.cuda.set_stream(self.torch.cuda.Stream(device=ctx.device, priority=-1))
        # ALWAYS bind the context to torch's current stream: the witness generators and the Python schedule mix torch ops (gathers,
        # concatenations, the caching allocator's stream-ordered reuse) with library kernels, and a second backend created in the same
        # process (another context on the same device) used to keep its context on the context's own stream — found by tools/stress_dist.py
        # as a wrong witness column in the first proof of a process with two contexts
        ctx.use_torch_stream()
        self.main = self.torch.cuda.current_stream(ctx.device)
        self.side = self.torch.cuda.Stream(device=ctx.device)

    @contextlib.contextmanager
    def overlap(self):
        """Work issued inside runs on a second HIP stream, ordered after everything already issued on the main
        stream.  Used for the coset NTTs, which only depend on finished columns and would otherwise wait behind
        an MSM whose tail occupies a few CUs.  join() makes the main stream wait for it."""
        ev = self.torch.cuda.Event()
        ev.record(self.main)
        self.side.wait_event(ev)
        with self.torch.cuda.stream(self.side):
            self.ctx.use_torch_stream()
            try:
                yield
            finally:
                pass
        self.ctx.use_torch_stream()

    def join(self):
        ev = self.torch.cuda.Event()
        ev.record(self.side)
        self.main.wait_event(ev)

    def setup(self, k, degree, s_int):
        # params_file (set by a caller that mirrors the reference's `gen_srs(k)` under PARAMS_DIR, /root/reference/src/bin/cli.rs:222):
        # read that SRS file if it is there (a real kzg_bn254_<k>.srs).  Otherwise generate the synthetic one (a PUBLIC trapdoor:
        # test / benchmark material only) and keep it beside it under a name the reference never reads, <stem>.synthetic.srs, so that
        # a later run of the reference CLI in the same directory cannot pick up an SRS whose trapdoor is known.
        pf = getattr(self, "params_file", None)
        syn = (pf[:-4] if pf and pf.endswith(".srs") else pf) + ".synthetic.srs" if pf else None
        src = pf if pf and os.path.exists(pf) else (syn if syn and os.path.exists(syn) else None)
        if src:
            self.params = self.ffi.ParamsKZG.read(self.ctx, src)
            if self.params.k != k:
                raise ValueError(f"{src}: k = {self.params.k}, expected {k}")
            self.params_source = src
        else:
            self.params = self.ffi.ParamsKZG.setup(self.ctx, k, self.fr(s_int))
            self.params_source = "generated"
            if syn and self.params.range()[1] == self.params.range()[2]:      # a point-range shard holds 1/N of the tables: nothing to write
                os.makedirs(os.path.dirname(syn) or ".", exist_ok=True)
                self.params.write(syn)
                self.params_source = syn
        self.domain = self.ffi.EvaluationDomain(self.ctx, degree, k)
        return self.domain

    def fr(self, x):
        """canonical int -> Montgomery limbs (host; uses the library's own host-side arithmetic)"""
        return fr_from_int_host(x)

    def fr_many(self, xs):
        return np.stack([fr_from_int_host(x) for x in xs]) if len(xs) else np.zeros((0, 4), dtype=np.uint64)

    def synth(self, n, seed):
        return self.ctx.synth_fill(n, seed)

    def synth_small(self, n, seed, bits_per_mille, word_bits):
        return self.ctx.synth_small(n, seed, bits_per_mille, word_bits)

    def gather(self, col, idx):
        """col[idx] (idx: host int64 array) — witness construction only"""
        return col[self.torch.from_numpy(idx).to(col.device)].contiguous()

    def put_rows(self, col, idx, vals):
        """col[idx] = vals in place (idx: host int64 array, distinct) — keygen-fixture construction only"""
        col[self.torch.from_numpy(idx).to(col.device)] = vals

    def from_host(self, arr):
        return self.ctx.to_device(arr)

    def concat(self, cols):
        return self.torch.cat(cols, dim=0)

    def coeff_to_lagrange(self, cols):
        self.domain.coeff_to_lagrange_device(cols)

    def permute(self, k, bf, cin, ctab, blind_in, blind_tab):
        return self.ffi.permute_expression_pair_device(self.ctx, k, bf, cin, ctab, blind_in, blind_tab)

    def lincomb(self, polys, coeffs, low):
        """sum_j coeffs[j] polys[j] - low (coeffs, low: canonical ints)"""
        return self.ffi.linear_combination_device(self.ctx, polys, self.fr_many(coeffs), self.fr_many(low) if low else None)

    def divide_by_linear(self, srcs, roots):
        """-> new polynomials srcs[j] / (X - roots[j])"""
        return self.ffi.divide_by_linear_device(self.ctx, srcs, self.fr_many(roots))

    def multiopen(self, polys, queries, flat_evals, squeeze, write_points):
        """ProverSHPLONK::create_proof inside the library (zkhip_shplonk_open); the transcript stays with the caller.
        queries: [(key, point int)]; flat_evals: their evaluations as (nq, 4) ABI rows."""
        keys = list(polys)
        index = {key: i for i, key in enumerate(keys)}
        mont = {}
        for _, pt in queries:
            if pt not in mont:
                mont[pt] = fr_from_int_host(pt)
        got = {}
        tags_sq = iter(("shplonk_y", "shplonk_v", "shplonk_u"))
        tags_wp = iter(("shplonk_h1", "shplonk_h2"))

        def sq():
            tag = next(tags_sq)
            got[tag] = squeeze(tag)
            return fr_from_int_host(got[tag])

        def wp(byts, xy):
            tag = next(tags_wp)
            got[tag] = (xy, byts)
            write_points(tag, [(xy, byts)])

        self.ffi.shplonk_open(self.ctx, self.params, [polys[key] for key in keys], [index[key] for key, _ in queries],
                              np.stack([mont[pt] for _, pt in queries]), flat_evals, wp, sq)
        return dict(y=got["shplonk_y"], v=got["shplonk_v"], u=got["shplonk_u"], h1=got["shplonk_h1"], h2=got["shplonk_h2"])

    def kate_division(self, polys, roots):
        """in place: polys[j] /= prod (X - r), r in roots[j] (canonical ints)"""
        self.ffi.kate_division_device(self.ctx, polys, [self.fr_many(r) for r in roots])

    def clone(self, cols):
        return [c.clone() for c in cols]

    def partial_commit(self, cols, lagrange, first, count):
        """Jacobian sums over the point range [first, first+count): (ncols, 12) int64 device tensor."""
        return self.params.commit_batch_device(cols, lagrange=lagrange, n=count, first=first)

    def g1_add(self, a, b):
        return self.ffi.g1_add(a, b)

    def finish(self, jac_rows):
        """host (ncols, 12) uint64 -> list of (affine (8,), 32 compressed bytes)"""
        aff = self.ffi.g1_batch_to_affine(jac_rows)
        return [(aff[j], self.ffi.g1_to_bytes(aff[j])) for j in range(aff.shape[0])]

    def commit_begin(self, cols, lagrange):
        """launches the MSMs of a batch; commit_end waits for them.  Host work placed between the two overlaps the GPU."""
        return self.partial_commit(cols, lagrange, 0, cols[0].shape[0]) if cols else None

    def commit_end(self, token):
        return self.ffi.commitments_read(self.ctx, token) if token is not None else []

    def commit(self, cols, lagrange):
        """one host round trip per batch (the Fiat-Shamir sync point)"""
        return self.commit_end(self.commit_begin(cols, lagrange))

    def lagrange_to_coeff(self, cols):
        self.domain.lagrange_to_coeff_device(cols)

    def coeff_to_extended(self, cols):
        return self.domain.coeff_to_extended_device(cols)

    def evaluate_h(self, kw):
        pack = self.ffi.EvalhPack()
        pack.build(**kw)
        return self.ffi.evaluate_h(self.ctx, pack, self.domain.extended_n)

    def evaluate_h_rows(self, kw, first_row, n_rows):
        """rows [first_row, first_row + n_rows) of evaluate_h: one rank's share of a row-sharded sweep"""
        pack = self.ffi.EvalhPack()
        pack.build(**kw)
        return self.ffi.evaluate_h_rows(self.ctx, pack, first_row, n_rows)

    def divide_and_to_coeff(self, h):
        self.domain.divide_by_vanishing_poly_device(h)
        self.domain.extended_to_coeff_device([h])
        return h

    def split(self, h, n, pieces):
        return [h[i * n:(i + 1) * n] for i in range(pieces)]

    def to_host(self, col):
        return self.ctx.to_host(col)

    def compress(self, graph, fixed_l, advice_l, instance_l, theta, k):
        """lookup::prover::compress_expressions: the graph (Horner in theta over the expressions) evaluated on the
        Lagrange domain itself — the sweep interpreter with rotation scale 1 (k = extended_k)."""
        zero = np.zeros(4, dtype=np.uint64)
        kw = dict(k=k, extended_k=k, cs_degree=3, blinding_factors=0, extended_omega=zero, g_coset=zero, delta=zero, beta=zero,
                  gamma=zero, theta=self.fr(theta), y=zero, fixed=fixed_l, advice=advice_l, instance=instance_l, challenges=[],
                  l0=None, l_last=None, l_active=None, gates_graph=graph, perm_columns=[], sigma=[], perm_z=[], lookup_graphs=[],
                  lookup_z=[], lookup_a=[], lookup_s=[], to_mont=self.fr_many)
        pack = self.ffi.EvalhPack()
        pack.build(**kw)
        return self.ffi.evaluate_h(self.ctx, pack, 1 << k)

    def permutation_products(self, k, values, sigmas, chunk_len, beta, gamma, bf, blinding):
        return self.ffi.permutation_products_device(self.ctx, k, values, sigmas, chunk_len, self.fr(beta), self.fr(gamma), bf, blinding)

    def lookup_product(self, k, cin, ctab, pin, ptab, beta, gamma, bf, blinding):
        return self.ffi.lookup_product_device(self.ctx, k, cin, ctab, pin, ptab, self.fr(beta), self.fr(gamma), bf, blinding)

    def eval_polys_at(self, polys, xs):
        """-> host (npolys, 4) ABI values polys[j](xs[j]); one launch pair and one transfer for the whole query set"""
        return self.ctx.to_host(self.ffi.eval_polynomials_at_device(self.ctx, polys, self.fr_many(xs)))

    def grand_products(self, k, beta, gamma, bf, values, sigmas, chunk_len, perm_blinding, lookups, lookup_blinding):
        return self.ffi.grand_products_device(self.ctx, k, self.fr(beta), self.fr(gamma), bf, values, sigmas, chunk_len, perm_blinding,
                                              lookups, lookup_blinding)

    def to_host_many(self, tensors):
        """one device->host transfer for a list of small results"""
        if not tensors:
            return []
        flat = self.ctx.to_host(self.torch.cat(tensors, dim=0))
        out, o = [], 0
        for t in tensors:
            out.append(flat[o:o + t.shape[0]])
            o += t.shape[0]
        return out

    def l_cosets(self, blinding_factors):
        """l_0, l_last, l_active_row cosets (keygen-time, plonk/keygen.rs) computed with the library's own NTTs."""
        n, dom = self.domain.n, self.domain
        one = fr_from_int_host(1)

        def unit(rows):
            a = np.zeros((n, 4), dtype=np.uint64)
            for r in rows:
                a[r] = one
            t = self.ctx.to_device(a)
            dom.lagrange_to_coeff_device([t])
            return dom.coeff_to_extended_device([t])[0]

        l0 = unit([0])
        l_blind = unit([n - 1 - i for i in range(blinding_factors)])
        l_last = unit([n - blinding_factors - 1])
        # l_active = 1 - (l_last + l_blind): evaluate with a one-op sweep so no extra kernel is needed
        g = ev.GraphEvaluator()
        s = g.add_calculation(ev.OP_ADD, [(ev.VS_FIXED, 0, g.add_rotation(0)), (ev.VS_FIXED, 1, 0)])
        g.add_calculation(ev.OP_SUB, [(ev.VS_CONSTANT, 1, 0), s])
        zero = np.zeros(4, dtype=np.uint64)
        kw = dict(k=dom.k, extended_k=dom.extended_k, cs_degree=3, blinding_factors=blinding_factors,
                  extended_omega=dom.extended_omega, g_coset=dom.g_coset, delta=zero, beta=zero, gamma=zero, theta=zero, y=zero,
                  fixed=[l_last, l_blind], advice=[], instance=[], challenges=[], l0=None, l_last=None, l_active=None,
                  gates_graph=g, perm_columns=[], sigma=[], perm_z=[], lookup_graphs=[], lookup_z=[], lookup_a=[], lookup_s=[],
                  to_mont=self.fr_many)
        l_active = self.evaluate_h(kw)
        return l0, l_last, l_active


class ShardedCommit:
    """Point-range sharding of every MSM over the ranks of a torch.distributed group:
    rank r sums points [r n/N, (r+1) n/N) of every column, the N x ncols partial sums (96 B each) are
    all-gathered as raw bytes and folded locally — RCCL has no curve-point reduction, and the payload is
    latency-sized.  With shard_ntt the coset NTTs are distributed by polynomial: rank r transforms
    columns r, r+N, ... of a batch and the results are all-gathered (512 MiB per extended column at k = 22 over xGMI against
    3.4 ms of transform: worth it from ~4 ranks; off by default).  Everything else is delegated to the wrapped backend unchanged
    (replicated)."""

    def __init__(self, inner, rank, world, dist, shard_ntt=False, shard_sweep=False):
        self.inner, self.rank, self.world, self.dist, self.shard_ntt = inner, rank, world, dist, shard_ntt
        self.shard_sweep = shard_sweep

    def evaluate_h(self, kw):
        """Row-sharded quotient sweep (the replicated-columns variant): every rank holds the complete
        extended columns, evaluates the rows [r e n / N, (r+1) e n / N) — rotations simply read across the boundary — and the
        N slices of h are all-gathered (32 B per row: 512 MiB at k = 22)."""
        en = self.inner.domain.extended_n
        if not self.shard_sweep or self.world == 1 or en % (64 * self.world):
            return self.inner.evaluate_h(kw)
        import torch

        rows = en // self.world
        mine = self.inner.evaluate_h_rows(kw, self.rank * rows, rows)
        is_t = torch.is_tensor(mine)
        send = mine if is_t else torch.from_numpy(np.ascontiguousarray(mine).view(np.int64))
        recv = [torch.empty_like(send) for _ in range(self.world)]
        self.dist.all_gather(recv, send)
        full = torch.cat(recv, dim=0)
        return full if is_t else full.numpy().view(np.uint64)

    def coeff_to_extended(self, cols):
        if not self.shard_ntt or self.world == 1 or not cols:
            return self.inner.coeff_to_extended(cols)
        import torch

        mine = self.inner.coeff_to_extended(cols[self.rank::self.world])
        as_t = lambda a: a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a).view(np.int64))
        back = (lambda t: t) if (mine and torch.is_tensor(mine[0])) or (not mine and torch.is_tensor(cols[0])) else \
            (lambda t: t.numpy().view(np.uint64))
        shape_src = as_t(mine[0]) if mine else None
        out = [None] * len(cols)
        rounds = -(-len(cols) // self.world)
        for t_ in range(rounds):
            have = t_ < len(mine)
            if shape_src is None:      # this rank owns no column of the batch: learn the shape from the extended size
                en = self.inner.domain.extended_n
                ref = as_t(cols[0])
                shape_src = ref.new_empty((en, 4))
            send = as_t(mine[t_]) if have else torch.zeros_like(shape_src)
            recv = [torch.empty_like(send) for _ in range(self.world)]
            self.dist.all_gather(recv, send)
            for r in range(self.world):
                j = t_ * self.world + r
                if j < len(cols):
                    out[j] = back(recv[r]) if r != self.rank or not have else mine[t_]
        return out

    def __getattr__(self, name):
        if name == "multiopen":      # the library's one-call multi-open commits unsharded: use the host-side prover over commit()
            raise AttributeError(name)
        return getattr(self.inner, name)

    def commit(self, cols, lagrange):
        return self.commit_end(self.commit_begin(cols, lagrange))

    def commit_begin(self, cols, lagrange):
        if not cols:
            return None
        n = cols[0].shape[0]
        lo, hi = self.rank * n // self.world, (self.rank + 1) * n // self.world
        return (self.inner.partial_commit(cols, lagrange, lo, hi - lo), len(cols))

    def commit_end(self, token):
        if token is None:
            return []
        import torch

        part, ncols = token
        if not torch.is_tensor(part):
            part = torch.from_numpy(np.ascontiguousarray(part).view(np.int64))
        outs = [torch.empty_like(part) for _ in range(self.world)]
        self.dist.all_gather(outs, part)
        parts = [o.cpu().numpy().view(np.uint64).reshape(ncols, 12) for o in outs]
        total = parts[0].copy()
        for p in parts[1:]:
            for j in range(ncols):
                total[j] = self.inner.g1_add(total[j], p[j])
        return self.inner.finish(total)


def from_mont_host(limbs):
    """Montgomery limbs (4 x u64) -> canonical int"""
    v = sum(int(limbs[i]) << (64 * i) for i in range(4))
    return v * R_INV_256 % R


def eval_ints(trace):
    """(key, rotation) -> canonical int for every evaluation of a proof trace"""
    return {q_: from_mont_host(row) for q_, row in trace["evals"]}


def splitmix64(x):
    """numpy uint64 -> uint64 (the finaliser of oracle/pyref.py's PRNG spec; used for the lookup witness' row choice)"""
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def fr_from_int_host(x):
    """canonical int -> Montgomery (x * 2^256 mod r) limbs; plain Python big-int arithmetic."""
    v = (x % R) * (1 << 256) % R
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


ROOT_OF_UNITY = 0x03DDB9F5166D18B798865EA93DD31F743215CF6DD39329C8D34F1ED960C37C9C   # order 2^28 (SURVEY.md §8 constants)
ZETA = 0xB3C4D79D41A917585BFC41088D8DAAA78B17EA66B99C90DD
DELTA = 0x09226B6E22C6F0CA64EC26AAD4C86E715B5F898E5E963F25870E56BBE533E9A2


class Prover:
    """Keygen-shaped setup once, then prove() = one create_proof-shaped pass (the benchmark step)."""
    _key_counter = 0

    def __init__(self, backend, shape, srs_trapdoor=0x1D5C0FFEE, satisfiable=False, key_file=None):
        """satisfiable=True (halo2-lib shaped circuits only): selectors, copy constraints and witness are built so that every
        gate, the permutation and the lookup hold, i.e. prove() returns a proof the verifier accepts.  The arithmetic is the
        same either way; with False the fixed / sigma columns are uniform synthetic values.
        key_file: a formats.ProvingKeyFile (what `read_pk` returns in the reference, /root/reference/src/bin/cli.rs:312,335,362,455,509):
        the key's fixed / sigma columns in all their forms and the l-polynomials come from the FILE — nothing of the key is generated —
        and the witness is the caller's (load_witness)."""
        self.b, self.shape = backend, shape
        self.satisfiable = satisfiable
        self.dom = backend.setup(shape.k, shape.degree, srs_trapdoor)
        self.n = 1 << shape.k
        sh, b, n = shape, backend, self.n
        seed = sh.seed * 1000
        self._ext = None
        if key_file is not None:
            if key_file.k != sh.k or len(key_file.fixed_values) != sh.n_fixed or len(key_file.permutations) != len(sh.perm_columns):
                raise ValueError(f"proving key file (k = {key_file.k}, {len(key_file.fixed_values)} fixed, {len(key_file.permutations)} permutation "
                                 f"columns) does not match the circuit shape {sh.name} (k = {sh.k}, {sh.n_fixed}, {len(sh.perm_columns)})")
            up = lambda cols: [b.from_host(np.array(c, dtype=np.uint64)) for c in cols]      # a copy: the file is a read-only mapping
            self.fixed_lagrange, self.fixed_coeff = up(key_file.fixed_values), up(key_file.fixed_polys)
            self.sigma_lagrange, self.sigma_coeff = up(key_file.permutations), up(key_file.permutation_polys)
            if key_file.extended_k == self.dom.extended_k:
                # the file's extended forms are used as they are (uploaded on first use: the coset-quotient path never touches them)
                self._ext_file = key_file
            self.key_source = "file"
        else:
            # pk: fixed columns and sigma polynomials in Lagrange form, coefficient form and as extended cosets; l cosets
            self.fixed_lagrange = [b.synth(n, seed + 100 + i) for i in range(sh.n_fixed)]
            self.sigma_lagrange = [b.synth(n, seed + 200 + i) for i in range(len(sh.perm_columns))]
            if satisfiable:
                self._build_satisfiable(seed)
            self.fixed_coeff = b.clone(self.fixed_lagrange)
            b.lagrange_to_coeff(self.fixed_coeff)
            self.sigma_coeff = b.clone(self.sigma_lagrange)
            b.lagrange_to_coeff(self.sigma_coeff)
            self.key_source = "generated"
        # The extended-domain forms of the key (fixed / sigma cosets, l_0 / l_last / l_active) are built on first use: the Python
        # schedule needs them, zkhip_create_proof_ex does not when it evaluates the quotient on cosets of the size-n domain
        # (cs_degree - 1 < extension factor: csrc/cosets.hip derives the key's columns in its own layout from the coefficient forms).
        self.gates_graph = ev.build_custom_gates(sh.gates)
        self.lookup_graphs = [ev.build_lookup(i, t) for i, t in sh.lookups]
        self.compress_graphs = []
        for inputs, tables in sh.lookups:
            pair = []
            for exprs in (inputs, tables):
                g = ev.GraphEvaluator()
                parts = [g.add_expression(x) for x in exprs]
                g.add_calculation(ev.OP_HORNER, [(ev.VS_CONSTANT, 0, 0), (ev.VS_THETA, 0, 0)] + parts)
                pair.append(g)
            self.compress_graphs.append(pair)
        self.omega = pow(ROOT_OF_UNITY, 1 << (28 - sh.k), R)
        # vk.transcript_repr stand-in (upstream: a Blake2b hash of the verifying key's Debug form, absorbed first by vk.hash_into)
        self.vk_repr = fr_from_int_host(int.from_bytes(hashlib.blake2b(sh.name.encode(), digest_size=64).digest(), "little") % R)

    @classmethod
    def from_explicit(cls, backend, shape, fixed_columns, copy_cells, srs_trapdoor=0x1D5C0FFEE, vk_repr=None):
        """A proving key from EXPLICIT data instead of the synthetic keygen: fixed_columns = one (n, 4) uint64 host array (ABI form) per
        fixed column of the shape; copy_cells = [(permutation column, row, permutation column, row)] two-cycles over shape.perm_columns;
        vk_repr = the verifying key's transcript representation (ABI limbs) or None for the stand-in.  What a circuit written against
        upstream's API hands to keygen — the consumer of tests/golden/reference_vectors.json's `prover` section builds its key this way."""
        p = cls(backend, shape, srs_trapdoor=srs_trapdoor, satisfiable=False)
        b, n = backend, 1 << shape.k
        if len(fixed_columns) != shape.n_fixed:
            raise ValueError(f"{len(fixed_columns)} fixed columns for a shape with {shape.n_fixed}")
        p.fixed_lagrange = [b.from_host(np.ascontiguousarray(c, dtype=np.uint64).reshape(n, 4)) for c in fixed_columns]
        p._set_copy_constraints([(ca * n + ra, cb * n + rb) for ca, ra, cb, rb in copy_cells])
        p.fixed_coeff = b.clone(p.fixed_lagrange)
        b.lagrange_to_coeff(p.fixed_coeff)
        p.sigma_coeff = b.clone(p.sigma_lagrange)
        b.lagrange_to_coeff(p.sigma_coeff)
        p.key_source = "explicit"
        if vk_repr is not None:
            p.vk_repr = np.ascontiguousarray(vk_repr, dtype=np.uint64).reshape(4)
        return p

    def _extended_key(self):
        if self._ext is None and getattr(self, "_ext_file", None) is not None:
            kf, b = self._ext_file, self.b
            up = lambda cols: [b.from_host(np.array(c, dtype=np.uint64)) for c in cols]
            self._ext = dict(fixed=up(kf.fixed_cosets), sigma=up(kf.permutation_cosets), l0=up([kf.l0])[0], l_last=up([kf.l_last])[0],
                             l_active=up([kf.l_active_row])[0])
        if self._ext is None:
            b = self.b
            l0, l_last, l_active = b.l_cosets(self.shape.blinding_factors)
            self._ext = dict(fixed=b.coeff_to_extended(self.fixed_coeff), sigma=b.coeff_to_extended(self.sigma_coeff), l0=l0, l_last=l_last,
                             l_active=l_active)
        return self._ext

    fixed_cosets = property(lambda self: self._extended_key()["fixed"])
    sigma_cosets = property(lambda self: self._extended_key()["sigma"])
    l0 = property(lambda self: self._extended_key()["l0"])
    l_last = property(lambda self: self._extended_key()["l_last"])
    l_active = property(lambda self: self._extended_key()["l_active"])

    @property
    def coset_quotient_applies(self):
        """cs_degree - 1 cosets of the size-n domain are fewer rows than the extended domain (the library's default path then)"""
        return self.dom.quotient_poly_degree < (1 << (self.dom.extended_k - self.shape.k))

    @property
    def n_instance_values(self):
        """values per instance column (the RSA circuit exposes 32 digest bytes, /root/reference/src/helpers.rs:167; the rest of the
        column is zero as in upstream's instance polynomial)"""
        return min(32, self.n - (self.shape.blinding_factors + 1))

    def _build_satisfiable(self, seed):
        """Selectors, copy constraints and the witness recipe of a satisfiable instance (witness-synthesis stand-in):
          * selector of basic column c = 1 on rows 0, 4, 8, ... (the vertical gate a + b c - d over rows i .. i+3);
          * copy constraints: usable/8 two-cycles, each tying the `b` input of a gate to a cell of the lookup-advice, constants,
            instance or another basic column (`c` inputs) — sigma is the identity sigma_j(w^i) = delta^j w^i with those swapped;
          * witness(): free cells random, copies gathered, lookup inputs drawn from the table, gate outputs computed by the
            sweep interpreter on the Lagrange domain.
        Only index arithmetic happens on the host; every field value is produced by the backend."""
        sh, b, n = self.shape, self.b, self.n
        if sh.layout == "sha":
            return self._build_satisfiable_sha(seed)
        assert sh.n_basic and sh.gates and len(sh.gates) == sh.n_basic + sh.n_phase1, "satisfiable instances need the halo2-lib gate shape"
        u = n - (sh.blinding_factors + 1)
        one = fr_from_int_host(1)
        gate_rows = np.arange(0, u - 3, 4)
        sel = np.zeros((n, 4), dtype=np.uint64)
        sel[0:len(gate_rows) * 4:4] = one                # rows gate_rows = 0, 4, 8, ... (a slice, not an index array)
        out_mask = np.zeros((n, 4), dtype=np.uint64)
        out_mask[3:len(gate_rows) * 4:4] = one           # rows gate_rows + 3
        for c in range(sh.n_basic):
            self.fixed_lagrange[c] = b.from_host(sel)
        self._out_mask = b.from_host(out_mask)
        # copy pairs (global cell id = perm-column index * n + row)
        P = len(sh.perm_columns)
        col_of = {pc: j for j, pc in enumerate(sh.perm_columns)}
        rng = np.random.default_rng(seed)
        partner_cols = [col_of[("advice", sh.n_basic + l)] for l in range(sh.n_lookup)] + [col_of[("fixed", sh.n_basic)]] \
            + [col_of[("instance", i)] for i in range(sh.n_instance)] + [col_of[("advice", c)] for c in range(sh.n_basic)]
        m = max(1, u // 8)
        src_rows = rng.permutation(gate_rows)[:m] + 1                      # `b` inputs
        m = len(src_rows)
        dst_rows = rng.permutation(gate_rows)[:m] + 2                      # `c` inputs when the partner is a basic column
        other_rows = rng.permutation(u)[:m]
        # pair t ties the `b` input of a gate in basic column t mod n_basic to a cell of partner column partner_cols[t mod len]: a `c` input
        # if the partner is a basic column, any usable row otherwise; an instance partner takes the column's public inputs in turn (one copy
        # per public input) and falls back to the constants column once they are used up.  (numpy throughout: m = usable / 8 pairs.)
        t_ = np.arange(m)
        pc = np.asarray(partner_cols, dtype=np.int64)[t_ % len(partner_cols)]
        is_basic = np.array([kind == "advice" and idx < sh.n_basic for kind, idx in sh.perm_columns])
        prow = np.where(is_basic[pc], dst_rows, other_rows).astype(np.int64)
        for i in range(sh.n_instance):
            hit = pc == col_of[("instance", i)]
            turn = np.cumsum(hit) - 1
            take = hit & (turn < self.n_instance_values)
            prow[take] = turn[take]
            pc[hit & ~take] = col_of[("fixed", sh.n_basic)]
        adv_col = np.asarray([col_of[("advice", c)] for c in range(sh.n_basic)], dtype=np.int64)
        pairs = np.stack([adv_col[t_ % sh.n_basic] * n + src_rows.astype(np.int64), pc * n + prow], axis=1)
        self._set_copy_constraints(pairs)
        # out = mask * (a(-3) + a(-2) a(-1)) + (1 - mask) * a(0), per basic column; the mask is appended to the fixed columns
        self._fill_graphs = []
        A = lambda c, r: ("advice", c, r)
        for c in range(sh.n_basic):
            mk = ("fixed", sh.n_fixed, 0)
            e = ("sum", ("prod", mk, ("sum", A(c, -3), ("prod", A(c, -2), A(c, -1)))), ("prod", ("sum", ("const", 1), ("neg", mk)), A(c, 0)))
            self._fill_graphs.append((c, self._fill_graph(e)))

    @staticmethod
    def _fill_graph(e):
        g = ev.GraphEvaluator()
        r_ = g.add_expression(e)
        g.add_calculation(ev.OP_HORNER, [(ev.VS_CONSTANT, 0, 0), (ev.VS_THETA, 0, 0), r_])
        return g

    def _set_copy_constraints(self, pairs):
        """sigma = the identity permutation sigma_j(w^i) = delta^j w^i with the cells of every pair swapped (two-cycles);
        _value_src[cell] = the cell a copy takes its value from (pairs are (destination, source))"""
        sh, b, n = self.shape, self.b, self.n
        P = len(sh.perm_columns)
        one = fr_from_int_host(1)
        pairs = np.asarray(pairs, dtype=np.int64).reshape(-1, 2)
        assert len(np.unique(pairs)) == pairs.size, "copy pairs must be disjoint"
        # witness(): the destination cells of permutation column j take their values from these source cells —
        # _value_src[j] = [(source permutation column, destination rows, source rows)]
        def by_columns(cells):
            """(destination cell, source cell) rows -> [(destination column, source column, destination rows, source rows)], one entry per
            column pair that occurs: ONE stable radix sort of a small key instead of P^2 boolean masks over all pairs"""
            if not len(cells):
                return []
            dst, src = np.ascontiguousarray(cells[:, 0]), np.ascontiguousarray(cells[:, 1])
            dcol, drow, scol, srow = dst >> sh.k, dst & (n - 1), src >> sh.k, src & (n - 1)      # n = 2^k
            key = (dcol * P + scol).astype(np.uint16)
            order = np.argsort(key, kind="stable")
            counts = np.bincount(key, minlength=P * P)
            out, at = [], 0
            for v in np.flatnonzero(counts):
                idx = order[at:at + counts[v]]
                at += counts[v]
                out.append((int(v) // P, int(v) % P, np.ascontiguousarray(drow[idx]), np.ascontiguousarray(srow[idx])))
            return out

        self._value_src = {}
        for dc, sc, rows_d, rows_s in by_columns(pairs):
            self._value_src.setdefault(dc, []).append((sc, rows_d, rows_s))
        self.copy_pairs = pairs
        # identity permutation values delta^j * w^i, then the swaps
        xpoly = np.zeros((n, 4), dtype=np.uint64)
        xpoly[1] = one
        omega_col = [b.from_host(xpoly)]
        b.coeff_to_lagrange(omega_col)
        # sigma = the identity columns with the cells of every pair swapped.  Sparse: the values of the ~n / 4 cells that move are gathered
        # from the pristine identity columns first, then put in place (until round 4: one gather over all P n cells through a P n index
        # array and a concatenated copy of the identity — 1.7 GiB of temporaries at k = 22 for a permutation that moves 4 % of the cells)
        sigma = [b.lincomb(omega_col, [pow(DELTA, j, R)], None) for j in range(P)]
        both = np.concatenate([pairs, pairs[:, ::-1]], axis=0) if len(pairs) else pairs      # (destination cell, source cell), both directions
        moves = [(dc, rows_d, b.gather(sigma[sc], rows_s)) for dc, sc, rows_d, rows_s in by_columns(both)]
        for dc, rows_, vals_ in moves:
            b.put_rows(sigma[dc], rows_, vals_)
        self.sigma_lagrange = sigma

    def _build_satisfiable_sha(self, seed):
        """The SHA-256-bit-circuit shape (CircuitShape.sha256) as a satisfiable instance:
          * every fixed column (selectors q_* and "round constants" alike — the shape shares them) = the indicator of the ACTIVE rows:
            even rows in [4, usable - 4), so that a gate's rotations (-3 .. +3) never reach the blinding rows and the one recurrence
            gate (column c = 1 mod 4: a(c, r+1) = x + y + 4 x y z) does not chain from row to row;
          * columns c = 0 mod 4 hold bits, c = 1 mod 4 are free on even rows and take the recurrence value on the odd row after an
            active row, c = 2 mod 4 = the 7-term word recomposition of column c - 2, c = 3 mod 4 = -(a(c-1) + constant) — so every
            gate vanishes on every active row (and the degree-5 gate with its two selectors too);
          * copy constraints over the permutation columns (advice 0, advice 1, instance 0): each public input i is copied into a free
            (even-row) cell of advice 1, and usable/16 odd-row cells of advice 0 (unconstrained by the bit gate) are tied to other
            free cells of advice 1.
        Only index arithmetic happens on the host; every field value is produced by the backend."""
        sh, b, n = self.shape, self.b, self.n
        u = n - (sh.blinding_factors + 1)
        one = fr_from_int_host(1)
        active = np.arange(4, u - 4, 2)
        sel = np.zeros((n, 4), dtype=np.uint64)
        sel[active] = one
        for c in range(sh.n_fixed):
            self.fixed_lagrange[c] = b.from_host(sel)
        nxt = np.zeros((n, 4), dtype=np.uint64)
        nxt[active + 1] = one
        self._out_mask = b.from_host(nxt)
        col_of = {pc: j for j, pc in enumerate(sh.perm_columns)}
        a0, a1, ins = col_of[("advice", 0)], col_of[("advice", 1)], col_of[("instance", 0)]
        rng = np.random.default_rng(seed)
        free1 = rng.permutation(active)               # even rows of advice 1: free cells
        nv = min(self.n_instance_values, max(1, len(free1) // 2))   # public inputs that are copied into the circuit (tiny circuits: fewer)
        m = max(1, min(u // 16, len(free1) - nv))
        odd0 = rng.permutation(active + 1)[:m]        # odd rows of advice 0: not under the bit gate
        pairs = [(a1 * n + int(free1[i]), ins * n + i) for i in range(nv)]
        pairs += [(a0 * n + int(odd0[t]), a1 * n + int(free1[nv + t])) for t in range(m)]
        # the pairs (advice 0 <- advice 1) take their value from a cell that itself may be a copy of an instance value: order matters
        # in witness(): advice 1 is gathered first
        self._set_copy_constraints(pairs)
        A = lambda c, r: ("advice", c, r)
        mk = ("fixed", sh.n_fixed, 0)
        keep = ("sum", ("const", 1), ("neg", mk))
        self._fill_graphs = []
        for c in range(sh.n_basic):
            if c % 4 == 1:
                x, y, z = A(c, -1), A(c - 1, -1), A(c - 1, 0)
                val = ("sum", ("sum", x, y), ("scaled", ("prod", ("prod", x, y), z), 4))
                e = ("sum", ("prod", mk, val), ("prod", keep, A(c, 0)))
            elif c % 4 == 2:
                acc = A(c - 2, -3)
                for r in (-2, -1, 0, 1, 2, 3):
                    acc = ("sum", ("scaled", acc, 2), A(c - 2, r))
                e = acc
            elif c % 4 == 3:
                e = ("neg", ("sum", A(c - 1, 0), ("fixed", (c + 1) % sh.n_fixed, 0)))
            else:
                continue
            self._fill_graphs.append((c, self._fill_graph(e)))

    def witness(self, proof_seed, dist="uniform"):
        """Synthetic advice / instance tables in Lagrange form, resident on the device (untimed).  Lookup-advice columns
        take their values from the table column (about two occurrences of each of the first n/2 table rows), so the
        lookup argument is satisfiable like a real range check's.  dist: "uniform" = uniform field elements (the worst case for
        the commitments); "survey" = SURVEY.md 8(d)'s value mix: for the (unsatisfiable) SHA-256 shape 90 % bits and 10 % words < 2^32; for
        satisfiable halo2-lib instances the mix of the RSA / aggregation rows on the free cells (limbs, bits, uniform values)."""
        sh, n = self.shape, self.n
        base = sh.seed * 1000 + proof_seed * 100000
        if dist == "survey" and not self.satisfiable:
            advice = [self.b.synth_small(n, base + 1 + i, 900, 32) for i in range(sh.n_basic)]
        elif dist == "survey" and sh.layout == "halo2-lib":
            # SURVEY.md 8(d)'s value mix for the halo2-lib circuits (RSA: 70 % uniform, 20 % < 2^64, 10 % bits; aggregation: "as RSA but
            # 50 % < 2^88" -> 40 % uniform, 50 % 88-bit CRT limbs, 10 % bits) on the FREE cells; copies and gate outputs follow as usual
            limb_bits, p_small, p_bit = (88, 500, 100) if sh.name.startswith("agg") else (64, 200, 100)
            advice = []
            for i in range(sh.n_basic):
                b = self.b
                uni = b.synth(n, base + 1 + i)
                words = [b.synth_small(n, base + 400 + 8 * i + j, 0, min(32, limb_bits - 32 * j)) for j in range((limb_bits + 31) // 32)]
                small = b.lincomb(words, [1 << (32 * j) for j in range(len(words))], None)
                bits = b.synth_small(n, base + 460 + i, 1000, 1)
                pick = splitmix64(np.arange(n, dtype=np.uint64) + np.uint64(((base + 480 + i) << 32) & 0xFFFFFFFFFFFFFFFF)) % np.uint64(1000)
                choice = np.where(pick < p_bit, 2, np.where(pick < p_bit + p_small, 1, 0)).astype(np.int64)
                advice.append(b.gather(b.concat([uni, small, bits]), np.arange(n, dtype=np.int64) + choice * n))
        else:
            advice = [self.b.synth(n, base + 1 + i) for i in range(sh.n_basic)]
        for j in range(sh.n_lookup):
            idx = (splitmix64(np.arange(n, dtype=np.uint64) + np.uint64(((base + 20 + j) << 32) & 0xFFFFFFFFFFFFFFFF)) % np.uint64(n // 2)).astype(np.int64)
            advice.append(self.b.gather(self.fixed_lagrange[sh.n_fixed - 1], idx))
        for _ in range(sh.n_phase1):      # columns of a later phase exist only once that phase's challenges do (advice_for_phase): zeros until then
            advice.append(self.b.lincomb([advice[0]], [0], None))
        # instance columns: n_instance_values public inputs, zero-padded (upstream builds the instance polynomial the same way)
        nv = self.n_instance_values
        inst_vals = [self.b.to_host(self.b.synth(nv, base + 50 + i)) for i in range(sh.n_instance)]
        instance = []
        for v in inst_vals:
            col = np.zeros((n, 4), dtype=np.uint64)
            col[:nv] = v
            instance.append(self.b.from_host(col))
        if self.satisfiable:
            b = self.b
            if sh.layout == "sha":
                for c in range(0, sh.n_basic, 4):      # the bit columns
                    advice[c] = b.synth_small(n, base + 1 + c, 1000, 1)
            copy_cols = [(j, i) for j, (t, i) in enumerate(sh.perm_columns) if t == "advice" and i < sh.n_basic]
            for j, c in (reversed(copy_cols) if sh.layout == "sha" else copy_cols):       # copies: basic-advice cells take their partner's value
                cols = {"advice": advice, "fixed": self.fixed_lagrange, "instance": instance}
                # (sparse: only the destination cells are touched; source cells are never destinations, so gathering before putting reads
                # what the full-column gather of earlier rounds read)
                moves = [(rows_d, b.gather(cols[sh.perm_columns[sc][0]][sh.perm_columns[sc][1]], rows_s)) for sc, rows_d, rows_s in self._value_src.get(j, [])]
                for rows_d, vals_ in moves:
                    b.put_rows(advice[c], rows_d, vals_)
            for c, g in self._fill_graphs:       # gate outputs
                advice[c] = b.compress(g, self.fixed_lagrange + [self._out_mask], advice, instance, 0, sh.k)
        return dict(advice=advice, instance=instance, instance_values=inst_vals, base=base)

    def advice_for_phase(self, wit, phase, user_challenges):
        """Witness synthesis of a later phase (the circuit's job upstream: `synthesize` runs again with the challenges of the earlier
        phases): fills wit["advice"][c] for the columns of `phase`.  user_challenges: canonical ints, index = challenge index."""
        sh = self.shape
        if wit.get("advice_for_phase") is not None:      # the caller's own synthesis (a witness that did not come from witness())
            wit["advice_for_phase"](wit, phase, user_challenges)
            wit.pop("advice_host", None)
            return
        for c in range(sh.n_advice):
            if sh.advice_phase[c] == phase and phase > 0:
                wit["advice"][c] = self.b.lincomb([wit["advice"][0]], [user_challenges[0]], None)
        wit.pop("advice_host", None)

    def _query_list(self):
        """[(key, rotation)] in upstream's query order (what create_proof evaluates at x and SHPLONK opens)"""
        sh = self.shape
        L, Zp = len(sh.lookups), sh.n_perm_sets
        last_rot = -(sh.blinding_factors + 1)
        qlist = [(("advice", col), rot) for kind, col, rot in sh.queries() if kind == "advice"]
        for i_ in range(Zp):
            qlist += [(("perm_z", i_), 0), (("perm_z", i_), 1)]
        for i_ in reversed(range(Zp - 1)):
            qlist.append((("perm_z", i_), last_rot))
        for i_ in range(L):
            qlist += [(("lookup_z", i_), 0), (("lookup_a", i_), 0), (("lookup_s", i_), 0), (("lookup_a", i_), -1), (("lookup_z", i_), 1)]
        qlist += [(("fixed", col), rot) for kind, col, rot in sh.queries() if kind == "fixed"]
        qlist += [(("sigma", i_), 0) for i_ in range(len(sh.perm_columns))]
        qlist += [(("h", 0), 0), (("random", 0), 0)]
        return qlist

    def _eval_write_order(self):
        """[(key, rotation)] in the order upstream WRITES the evaluations to the transcript (plonk/prover.rs after squeezing x
        [UPSTREAM-RECALL]): advice evals, fixed evals, vanishing's random_eval, the permutation's common (sigma) evals, per
        permutation set z(x), z(wx) and — all but the last set — z(w^last x), per lookup z(x), z(wx), a'(x), a'(w^-1 x), s'(x).
        h(x) is not written.  (The multi-open consumes the same evaluations in _query_list()'s order.)"""
        sh = self.shape
        L, Zp = len(sh.lookups), sh.n_perm_sets
        last_rot = -(sh.blinding_factors + 1)
        w = [(("advice", col), rot) for kind, col, rot in sh.queries() if kind == "advice"]
        w += [(("fixed", col), rot) for kind, col, rot in sh.queries() if kind == "fixed"]
        w.append((("random", 0), 0))
        w += [(("sigma", i_), 0) for i_ in range(len(sh.perm_columns))]
        for i_ in range(Zp):
            w += [(("perm_z", i_), 0), (("perm_z", i_), 1)]
            if i_ + 1 < Zp:
                w.append((("perm_z", i_), last_rot))
        for i_ in range(L):
            w += [(("lookup_z", i_), 0), (("lookup_z", i_), 1), (("lookup_a", i_), 0), (("lookup_a", i_), -1), (("lookup_s", i_), 0)]
        return w

    def save_witness(self, wit, path):
        """the advice columns and instance values of a witness as one .npz (host arrays, ABI form): what a circuit's witness generation
        hands to create_proof — the synthetic stand-in for `circuit` + `instances` of gen_snark_shplonk (/root/reference/src/helpers.rs:233)"""
        h = self.b.to_host
        np.savez(path, base=np.int64(wit["base"]), n_advice=len(wit["advice"]), n_instance=len(wit["instance_values"]),
                 **{f"advice_{i}": h(c) for i, c in enumerate(wit["advice"])},
                 **{f"instance_values_{i}": np.asarray(v, dtype=np.uint64) for i, v in enumerate(wit["instance_values"])})

    def load_witness(self, path):
        z = np.load(path)
        b, n = self.b, self.n
        advice = [b.from_host(z[f"advice_{i}"]) for i in range(int(z["n_advice"]))]
        inst_vals = [np.array(z[f"instance_values_{i}"], dtype=np.uint64).reshape(-1, 4) for i in range(int(z["n_instance"]))]
        if len(advice) != self.shape.n_advice or any(len(b.to_host(c)) != n for c in advice[:1]):
            raise ValueError(f"{path}: {len(advice)} advice columns for a circuit with {self.shape.n_advice}")
        instance = []
        for v in inst_vals:
            col = np.zeros((n, 4), dtype=np.uint64)
            col[:len(v)] = v
            instance.append(b.from_host(col))
        return dict(advice=advice, instance=instance, instance_values=inst_vals, base=int(z["base"]))

    def release(self):
        """drops what the library's context caches for this proving key (zkhip_key_release): call when the Prover is done"""
        if getattr(self, "_npk", None) is not None and hasattr(self.b, "ctx") and hasattr(self.b.ctx, "key_release"):
            self.b.ctx.key_release(self._npk.key_id)

    def _native_key(self, with_extended=False):
        """zk_proving_key for zkhip_create_proof (built once; the arrays it points to are kept alive on self)"""
        if getattr(self, "_npk", None) is not None and not (with_extended and not self._npk.fixed_cosets and not self._npk.l0):
            return self._npk
        import ctypes as C

        ffi, sh, b = self.b.ffi, self.shape, self.b
        keep = []

        def ptrs(tensors):
            arr = (C.c_void_p * max(1, len(tensors)))(*[t.data_ptr() for t in tensors])
            keep.append((arr, tensors))
            return C.cast(arr, C.c_void_p)

        def arr(values, dtype):
            a = np.array(list(values) or [0], dtype=dtype)
            keep.append(a)
            return a.ctypes.data

        pack = ffi.EvalhPack()
        keep.append(pack)

        def graphs(gs):
            ga = (ffi.ZkGraph * max(1, len(gs)))(*[pack.graph(g, b.fr_many) for g in gs])
            keep.append(ga)
            return C.cast(ga, C.c_void_p)

        pk = ffi.ZkProvingKey()
        pk.k, pk.cs_degree, pk.blinding_factors = sh.k, sh.degree, sh.blinding_factors
        pk.n_fixed, pk.n_advice, pk.n_instance = len(self.fixed_lagrange), sh.n_advice, sh.n_instance
        pk.n_lookups, pk.n_perm_columns = len(sh.lookups), len(sh.perm_columns)
        pk.g, pk.g_lagrange, pk.domain = b.params.g, b.params.g_lagrange, b.domain.h
        pk.fixed_lagrange, pk.fixed_coeff = ptrs(self.fixed_lagrange), ptrs(self.fixed_coeff)
        pk.sigma_lagrange, pk.sigma_coeff = ptrs(self.sigma_lagrange), ptrs(self.sigma_coeff)
        if self._ext is not None or not self.coset_quotient_applies or with_extended:
            pk.fixed_cosets, pk.sigma_cosets = ptrs(self.fixed_cosets), ptrs(self.sigma_cosets)
            pk.l0, pk.l_last, pk.l_active_row = self.l0.data_ptr(), self.l_last.data_ptr(), self.l_active.data_ptr()
        # else: NULL — the coset path derives them; prove_native retries with them if the library asks
        pk.custom_gates = pack.graph(self.gates_graph, b.fr_many)
        pk.lookup_graphs = graphs(self.lookup_graphs)
        pk.lookup_input_compress = graphs([p_[0] for p_ in self.compress_graphs])
        pk.lookup_table_compress = graphs([p_[1] for p_ in self.compress_graphs])
        single = lambda exprs, kind: exprs[0][1] if len(exprs) == 1 and exprs[0][0] == kind and exprs[0][2] == 0 else -1
        pk.lookup_input_advice_column = arr((single(ins_, "advice") for ins_, _ in sh.lookups), np.int32)
        pk.lookup_table_fixed_column = arr((single(tabs_, "fixed") for _, tabs_ in sh.lookups), np.int32)
        Prover._key_counter += 1
        pk.key_id = (os.getpid() << 32) | Prover._key_counter     # unique per proving key in this process: keys the sorted-table cache
        tmap = {"advice": 0, "fixed": 1, "instance": 2}
        pk.perm_column_type = arr((tmap[t] for t, _ in sh.perm_columns), np.uint32)
        pk.perm_column_index = arr((i for _, i in sh.perm_columns), np.uint32)
        aq = [(c, r) for kind, c, r in sh.queries() if kind == "advice"]
        fq = [(c, r) for kind, c, r in sh.queries() if kind == "fixed"]
        pk.n_advice_queries, pk.n_fixed_queries = len(aq), len(fq)
        pk.advice_query_column, pk.advice_query_rotation = arr((c for c, _ in aq), np.uint32), arr((r for _, r in aq), np.int32)
        pk.fixed_query_column, pk.fixed_query_rotation = arr((c for c, _ in fq), np.uint32), arr((r for _, r in fq), np.int32)
        pk.delta = (C.c_uint64 * 4)(*[int(v) for v in b.fr(DELTA)])
        pk.vk_transcript_repr = arr(self.vk_repr, np.uint64)
        if any(sh.advice_phase) or sh.challenge_phase:
            pk.advice_column_phase = arr(sh.advice_phase, np.uint8)
            pk.n_challenges = len(sh.challenge_phase)
            pk.challenge_phase = arr(sh.challenge_phase, np.uint8)
        self._npk, self._npk_keep = pk, keep
        return pk

    def prove_native(self, wit, fetch_h=False, python_transcript=False, evm=False, transcript=None, host_inputs=False, blinding=None,
                     auto_extended=True):
        """The same pass through zkhip_create_proof_ex (the schedule and all host arithmetic in the library).
        transcript: "blake2b" (halo2's Blake2bWrite, the default), "poseidon" (snark-verifier's native transcript: what the
        reference's prove-* commands and the aggregation snark use), "evm" (Keccak-256: gen-x509-agg-evm-proof; points are 64
        big-endian bytes in the proof) — all three inside the library, no Python between the launches — or, with python_transcript,
        Blake2bTranscript (hashlib) through callbacks (identical challenges to "blake2b": tests/test_schedule_cpu.py).
        host_inputs: the advice columns are handed over as pinned HOST arrays (what a Rust caller's Vec<Fr> columns are) and uploaded
        inside the call; the instance columns are built by the library from the instance values.
        blinding: None = the library's seeded generator; or dict(lookup_permuted, perm_z, lookup_z, random_poly) of (m, 4) uint64
        host arrays / device tensors — the caller's rng draws, as upstream's create_proof takes them from its `rng` argument.
        Returns the same trace as prove() plus trace["proof"], the bytes the transcript's writer received; the quotient's
        coefficients are copied to the host only on request (fetch_h: 96 n bytes over PCIe, for tests)."""
        import ctypes as C

        sh, b, n = self.shape, self.b, self.n
        ffi, ctx = b.ffi, b.ctx
        pk = self._native_key()
        L, Zp, A = len(sh.lookups), sh.n_perm_sets, sh.n_advice
        qd = self.dom.quotient_poly_degree
        kind = transcript or ("evm" if evm else "blake2b")
        point_tags = ["advice"] * A + ["lookup_permuted"] * (2 * L) + ["products"] * (Zp + L) + ["random_poly"] + ["quotient"] * qd \
            + ["shplonk_h1", "shplonk_h2"]
        user_order = [c_ for ph in sh.phases for c_, p_ in enumerate(sh.challenge_phase) if p_ == ph]      # squeeze order of the user challenges
        squeeze_tags = [("user", c_) for c_ in user_order] + ["theta", "beta", "gamma", "y", "x", "shplonk_y", "shplonk_v", "shplonk_u"]
        trace = {"commitments": [], "challenges": {}, "points": {}, "transcript": kind}
        if user_order:
            trace["challenges"]["user"] = [None] * len(user_order)

        def record_challenge(tag, value):
            if isinstance(tag, tuple):
                trace["challenges"]["user"][tag[1]] = value
            else:
                trace["challenges"][tag] = value
        nt = None
        keep = []
        if python_transcript:
            ts, state = Blake2bTranscript(), dict(p=0, s=0)

            def write_point(byts, xy):
                tag = point_tags[state["p"]]
                state["p"] += 1
                trace["commitments"].append((tag, byts.hex()))
                trace["points"].setdefault(tag, []).append(xy)
                ts.write_point(xy)

            def squeeze():
                tag = squeeze_tags[state["s"]]
                state["s"] += 1
                c = ts.squeeze()
                record_challenge(tag, c)
                return fr_from_int_host(c)

            t = ffi.make_transcript(write_point, squeeze, ts.write_scalar, ts.common_scalar)
            keep.append(t)
            t_ref = C.byref(t)
        else:
            nt = ffi.LibTranscript(kind)
            t_ref = nt.callbacks
        qlist = self._query_list()
        evals = np.zeros((len(qlist), 4), dtype=np.uint64)
        worder = np.zeros(len(qlist), dtype=np.uint32)
        out = ffi.ZkProofOut()
        out.evals, out.evals_cap, out.eval_write_order = evals.ctypes.data, len(qlist), worder.ctypes.data
        inp = ffi.ZkProofInputs()
        if host_inputs:
            # True: pinned host arrays; "pageable": plain host arrays — what a Rust caller's Vec<Fr> columns really are (the library registers large ones for the call)
            key_ = "advice_host_pageable" if host_inputs == "pageable" else "advice_host"
            host_adv = wit.get(key_)
            if host_adv is None:
                host_adv = wit[key_] = [c_.cpu() if host_inputs == "pageable" else c_.cpu().pin_memory() for c_ in wit["advice"]]
            adv = (C.c_void_p * max(1, A))(*[c_.data_ptr() for c_ in host_adv])
            inp.advice_on_host = 1
            inp.d_instance = None
        else:
            adv = (C.c_void_p * max(1, A))(*[c_.data_ptr() for c_ in wit["advice"]])
            ins = (C.c_void_p * max(1, sh.n_instance))(*[c_.data_ptr() for c_ in wit["instance"]])
            keep.append(ins)
            inp.d_instance = C.cast(ins, C.c_void_p)
        inp.advice = C.cast(adv, C.c_void_p)
        if any(sh.advice_phase):
            # later phases: the library calls back once the challenges of the earlier phases exist; the witness of that phase is synthesised
            # here (advice_for_phase) and its columns' pointers stored into the advice array the library reads
            def next_phase(_user, phase, ch_ptr, adv_ptr):
                try:
                    nch = len(sh.challenge_phase)
                    user = [from_mont_host(np.array([ch_ptr[4 * c_ + q_] for q_ in range(4)], dtype=np.uint64)) for c_ in range(nch)]
                    self.advice_for_phase(wit, phase, user)
                    for c_ in range(A):
                        if sh.advice_phase[c_] != phase:
                            continue
                        col = wit["advice"][c_]
                        if host_inputs:
                            col = col.cpu().pin_memory()
                            keep.append(col)
                        adv_ptr[c_] = col.data_ptr()
                    ctx.use_torch_stream()
                    self.b.torch.cuda.current_stream(ctx.device).synchronize()      # the columns are complete before the library reads them
                    return 0
                except Exception as e:   # noqa: BLE001 — an exception must not cross the C boundary
                    print(f"advice_for_phase({phase}) failed: {e}", file=sys.stderr)
                    return 1
            cb = ffi.ADVICE_PHASE_FN(next_phase)
            keep.append(cb)
            inp.advice_phase = cb
        ivals = [np.ascontiguousarray(v, dtype=np.uint64) for v in wit["instance_values"]]
        ivp = (C.c_void_p * max(1, len(ivals)))(*[v.ctypes.data for v in ivals])
        ivl = np.array([len(v) for v in ivals] or [0], dtype=np.uint32)
        keep += [ivals, ivp, ivl]
        inp.instance_values, inp.instance_len = C.cast(ivp, C.c_void_p), ivl.ctypes.data
        inp.blinding_seed = wit["base"]
        if blinding is not None:
            bl = ffi.ZkBlinding()
            on_host = isinstance(blinding["random_poly"], np.ndarray)
            for name in ("lookup_permuted", "perm_z", "lookup_z", "random_poly"):
                v = blinding.get(name)
                if v is None or len(v) == 0:
                    continue
                if on_host:
                    v = np.ascontiguousarray(v, dtype=np.uint64)
                    setattr(bl, name, v.ctypes.data)
                else:
                    setattr(bl, name, v.data_ptr())
                keep.append(v)
            bl.on_host = 1 if on_host else 0
            keep.append(bl)
            inp.blinding = C.cast(C.pointer(bl), C.c_void_p)
        ctx.use_torch_stream()
        t_call = time.perf_counter()
        rc = ffi.lib().zkhip_create_proof_ex(ctx.h, C.byref(pk), C.byref(inp), t_ref, C.byref(out))
        trace["native_call_s"] = time.perf_counter() - t_call      # the library call alone (the rest of this function is Python bookkeeping)
        if rc != 0 and auto_extended and b"extended cosets are missing" in ffi.lib().zkhip_last_error():
            # the context runs the extended-domain path (coset_quotient = 0): hand over the key's extended forms (nothing has entered the
            # transcript yet: the library checks its inputs first)
            pk = self._native_key(with_extended=True)
            rc = ffi.lib().zkhip_create_proof_ex(ctx.h, C.byref(pk), C.byref(inp), t_ref, C.byref(out))
        if rc == ffi.ECONSTRAINT:
            raise ffi.ConstraintSystemFailure(ffi.lib().zkhip_last_error().decode())
        if rc != 0:
            raise ffi.ZkhipError(f"zkhip_create_proof: {rc}: {ffi.lib().zkhip_last_error().decode()}")
        n_eval_written = len(qlist) - 1
        if nt is not None:   # rebuild the trace from what the library's transcript recorded
            proof, chs, pts = nt.proof(), nt.challenges(), nt.points()
            trace["proof"] = proof
            psize = nt.POINT_BYTES[kind]
            assert len(chs) == len(squeeze_tags) and len(pts) == len(point_tags)
            off = 0
            for i_, tag in enumerate(point_tags):
                if tag == "shplonk_h1":
                    off += 32 * n_eval_written      # the evaluations sit between the quotient pieces and the SHPLONK points
                byts = proof[off:off + psize]
                off += psize
                trace["commitments"].append((tag, byts.hex()))
                trace["points"].setdefault(tag, []).append(pts[i_])
            assert off == len(proof)
            for tag, limbs in zip(squeeze_tags, chs):
                record_challenge(tag, from_mont_host(limbs))
        else:
            assert state["p"] == len(point_tags) and state["s"] == len(squeeze_tags)
        assert out.n_evals == len(qlist)
        trace["evals"] = [(q_, evals[i_]) for i_, q_ in enumerate(qlist)]
        trace["query_list"] = qlist
        trace["eval_write_order"] = [qlist[i_] for i_ in worder[:n_eval_written]]
        assert trace["eval_write_order"] == self._eval_write_order()
        trace["h_pieces"] = None
        if fetch_h:
            h = np.empty((qd * n, 4), dtype=np.uint64)
            ffi._check(ffi.lib().zkhip_memcpy_d2h(ctx.h, h.ctypes.data_as(C.c_void_p), C.c_void_p(out.d_h), C.c_size_t(qd * n * 32)))
            trace["h_pieces"] = [h[i_ * n:(i_ + 1) * n] for i_ in range(qd)]
        trace["opening"] = dict(y=trace["challenges"]["shplonk_y"], v=trace["challenges"]["shplonk_v"], u=trace["challenges"]["shplonk_u"])
        trace["n_commitments"] = len(trace["commitments"])
        return trace

    def prove(self, wit, transcript="blake2b-py", blinding=None):
        """One pass.  Returns the transcript trace: every commitment's bytes and the challenges.  transcript: "blake2b-py"
        (Blake2bTranscript above), or the library's "blake2b" / "evm" / "poseidon" driven from here (make_transcript).
        blinding: None = the seeded generator (wit["base"]); or the caller's rng draws as in prove_native: dict(lookup_permuted
        (2 L (bf + 1), 4), perm_z (Zp bf, 4), lookup_z (L bf, 4), random_poly (n, 4)) of uint64 host arrays."""
        sh, b, n, dom = self.shape, self.b, self.n, self.dom
        base = wit["base"]
        L, Zp = len(sh.lookups), sh.n_perm_sets
        bf_ = sh.blinding_factors
        if blinding is not None:
            hostb = {k_: np.ascontiguousarray(v_, dtype=np.uint64).reshape(-1, 4) for k_, v_ in blinding.items() if v_ is not None}
            draw = dict(random_poly=lambda: b.from_host(hostb["random_poly"]),
                        permuted=lambda j: b.from_host(hostb["lookup_permuted"][j * (bf_ + 1):(j + 1) * (bf_ + 1)]),
                        perm_z=lambda: b.from_host(hostb["perm_z"]) if Zp * bf_ else b.synth(0, 0),
                        lookup_z=lambda: b.from_host(hostb["lookup_z"]) if L * bf_ else b.synth(max(1, L) * bf_, 0))
        else:
            draw = dict(random_poly=lambda: b.synth(n, base + 380),
                        permuted=lambda j: b.synth(bf_ + 1, base + (300 if j % 2 == 0 else 320) + j // 2),
                        perm_z=lambda: b.synth(Zp * bf_, base + 340), lookup_z=lambda: b.synth(max(1, L) * bf_, base + 360))
        trace = {"commitments": [], "challenges": {}, "points": {}, "transcript": transcript}
        ts = make_transcript(transcript)
        # vk.hash_into(transcript), then every instance value (KZG: instances are hashed, not committed)
        ts.common_scalar(self.vk_repr)
        for col in wit["instance_values"]:
            for v in col:
                ts.common_scalar(v)

        def absorb(tag, pts):
            byts = [p[1] for p in pts]
            trace["points"].setdefault(tag, []).extend(p[0] for p in pts)
            trace["commitments"] += [(tag, x.hex()) for x in byts]
            for p_ in pts:
                ts.write_point(p_[0])
            return byts

        # 1. advice commitments (Lagrange basis).  The vanishing argument's random polynomial does not depend on
        #    any challenge, so its commitment (step 4, monomial basis) rides in the same MSM pass; it enters the
        #    transcript at its usual place.  The coset NTTs of finished columns (step 5) are issued on a second
        #    stream as soon as their inputs exist, so they run beside the latency-bound MSM tails.
        #    Phases (CircuitShape.advice_phase): the columns of phase p are committed and absorbed, then the user challenges of phase p are
        #    squeezed, then the witness of phase p + 1 is synthesised with them (advice_for_phase) — upstream's loop over `phases`.
        rand_poly = [draw["random_poly"]()]
        user_ch, rand_commit = [], None
        for ph in sh.phases:
            if ph > 0:
                self.advice_for_phase(wit, ph, user_ch)
            cols_ph = [i for i in range(sh.n_advice) if sh.advice_phase[i] == ph]
            batch = [wit["advice"][i] for i in cols_ph]
            if rand_commit is None:
                c1 = b.commit(batch + rand_poly, lagrange=[True] * len(batch) + [False])
                rand_commit = c1[len(batch):]
            else:
                c1 = b.commit(batch, lagrange=True)
            absorb("advice", c1[:len(batch)])
            user_ch += [ts.squeeze() for p_ in sh.challenge_phase if p_ == ph]
        advice = b.clone(wit["advice"])
        instance = b.clone(wit["instance"])
        with b.overlap():
            adv_coeff = b.clone(wit["advice"] + wit["instance"])
            b.lagrange_to_coeff(adv_coeff)
            ext_adv = b.coeff_to_extended(adv_coeff)
        theta = ts.squeeze()
        # 2. lookups: theta-compress the input / table expressions, permute_expression_pair (sort; blinding rows are
        #    seeded stand-ins for the rng); commit the permuted pair in coefficient form
        bf = sh.blinding_factors
        compressed = [(b.compress(gi, self.fixed_lagrange, wit["advice"], wit["instance"], theta, sh.k),
                       b.compress(gt, self.fixed_lagrange, wit["advice"], wit["instance"], theta, sh.k)) for gi, gt in self.compress_graphs]
        permuted = [b.permute(sh.k, bf, compressed[i][0], compressed[i][1], draw["permuted"](2 * i), draw["permuted"](2 * i + 1))
                    for i in range(L)]
        perm_in_l, perm_tab_l = [p_[0] for p_ in permuted], [p_[1] for p_ in permuted]
        perm_in, perm_tab = b.clone(perm_in_l), b.clone(perm_tab_l)
        b.lagrange_to_coeff(perm_in + perm_tab)
        with b.overlap():
            ext_perm = b.coeff_to_extended(perm_in + perm_tab)
        # transcript order: permuted input, permuted table, lookup by lookup (lookup::Argument::commit_permuted per lookup)
        inter = [c_ for pair in zip(perm_in, perm_tab) for c_ in pair]
        t2 = absorb("lookup_permuted", b.commit(inter, lagrange=False)) if L else []
        beta, gamma = ts.squeeze(), ts.squeeze()
        # 3. grand products: permutation (chunks of degree-2 columns) and one per lookup; blinding rows are seeded stand-ins
        cols = {"advice": wit["advice"], "fixed": self.fixed_lagrange, "instance": wit["instance"]}
        perm_values = [cols[t][i] for t, i in sh.perm_columns]
        perm_z, look_z = b.grand_products(sh.k, beta, gamma, bf, perm_values, self.sigma_lagrange, sh.degree - 2, draw["perm_z"](),
                                          [(compressed[i][0], compressed[i][1], perm_in_l[i], perm_tab_l[i]) for i in range(L)],
                                          draw["lookup_z"]())
        b.lagrange_to_coeff(perm_z + look_z)
        with b.overlap():
            ext_prod = b.coeff_to_extended(look_z + perm_z)
        t3 = absorb("products", b.commit(perm_z + look_z, lagrange=False))
        # 4. vanishing argument's random polynomial (committed in pass 1)
        t4 = absorb("random_poly", rand_commit)
        y = ts.squeeze()
        # 5. quotient: everything is on the extended coset by now; sweep, divide, back to coefficients
        b.join()
        adv_c, ins_c = ext_adv[:sh.n_advice], ext_adv[sh.n_advice:]
        pin_c, ptab_c = ext_perm[:L], ext_perm[L:]
        lz_c, pz_c = ext_prod[:L], ext_prod[L:]
        kw = dict(k=sh.k, extended_k=dom.extended_k, cs_degree=sh.degree, blinding_factors=sh.blinding_factors,
                  extended_omega=dom.extended_omega, g_coset=dom.g_coset, delta=b.fr(DELTA), beta=b.fr(beta), gamma=b.fr(gamma),
                  theta=b.fr(theta), y=b.fr(y), fixed=self.fixed_cosets, advice=adv_c, instance=ins_c, challenges=[b.fr(c_) for c_ in user_ch],
                  l0=self.l0, l_last=self.l_last, l_active=self.l_active, gates_graph=self.gates_graph,
                  perm_columns=sh.perm_columns, sigma=self.sigma_cosets, perm_z=pz_c, lookup_graphs=self.lookup_graphs,
                  lookup_z=lz_c, lookup_a=pin_c, lookup_s=ptab_c, to_mont=b.fr_many)
        h = b.evaluate_h(kw)
        h = b.divide_and_to_coeff(h)
        pieces = b.split(h, n, dom.quotient_poly_degree)
        quotient_commit = b.commit_begin(pieces, lagrange=False)
        # 5b. evaluations at x * omega^rot of everything the verifier queries (create_proof's eval_polynomial calls), in
        #     upstream's query order: advice, permutation products, lookups, fixed, sigma, vanishing (h, random poly).
        #     The query list does not depend on x: it is assembled while the quotient commitment is still running.
        polys = {}
        for i_, c_ in enumerate(adv_coeff[:sh.n_advice]):
            polys[("advice", i_)] = c_
        for i_, c_ in enumerate(self.fixed_coeff):
            polys[("fixed", i_)] = c_
        for i_, c_ in enumerate(self.sigma_coeff):
            polys[("sigma", i_)] = c_
        for i_, c_ in enumerate(perm_z):
            polys[("perm_z", i_)] = c_
        for i_ in range(L):
            polys[("lookup_z", i_)], polys[("lookup_a", i_)], polys[("lookup_s", i_)] = look_z[i_], perm_in[i_], perm_tab[i_]
        polys[("random", 0)] = rand_poly[0]
        qlist = self._query_list()
        t5 = absorb("quotient", b.commit_end(quotient_commit))
        x = ts.squeeze()
        xn = pow(x, n, R)
        polys[("h", 0)] = b.lincomb(pieces, [pow(xn, i_, R) for i_ in range(len(pieces))], None)   # sum_i x^(n i) h_i(X)
        rot_point = {rot: x * pow(self.omega, rot % n, R) % R for rot in {rot for _, rot in qlist}}
        points = [rot_point[rot] for _, rot in qlist]
        flat = b.eval_polys_at([polys[key] for key, _ in qlist], points)
        trace["evals"] = [(q_, flat[i_]) for i_, q_ in enumerate(qlist)]
        trace["query_list"] = qlist
        # the transcript receives every evaluation except h's (the verifier recomputes it), in upstream's WRITE order
        at = {q_: i_ for i_, q_ in enumerate(qlist)}
        trace["eval_write_order"] = self._eval_write_order()
        for q_ in trace["eval_write_order"]:
            ts.write_scalar(flat[at[q_]])
        # 6. SHPLONK multi-open of all of them: two more commitments
        if hasattr(b, "multiopen"):      # the library's ProverSHPLONK (host arithmetic in C++; evaluations stay in ABI form)
            opening = b.multiopen(polys, [(key, pt) for (key, _), pt in zip(qlist, points)], flat, lambda tag: ts.squeeze(), absorb)
        else:
            evals = eval_ints(trace)
            queries = [(key, pt, evals[(key, rot)]) for (key, rot), pt in zip(qlist, points)]
            opening = ShplonkProver(b).create_proof(polys, queries, lambda tag: ts.squeeze(), absorb)
        trace["challenges"] = dict(theta=theta, beta=beta, gamma=gamma, y=y, x=x, shplonk_y=opening["y"], shplonk_v=opening["v"],
                                   shplonk_u=opening["u"])
        if user_ch:
            trace["challenges"]["user"] = list(user_ch)      # the user challenges (multi-phase circuits only), index = challenge index
        trace["opening"] = opening
        trace["h_pieces"] = pieces
        trace["n_commitments"] = len(trace["commitments"])
        if hasattr(ts, "proof"):
            trace["proof"] = ts.proof()
        return trace
