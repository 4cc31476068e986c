"""On-disk artefacts the reference's CLI passes between its steps, as far as the proving path consumes them:

  * proving key  `gen_pk(.., Some(path))` / `read_pk::<C>(path, params)`   /root/reference/src/bin/cli.rs:247,268,294,312,335,362,402,455,509
    = halo2_proofs ProvingKey::write / ::read with SerdeFormat::RawBytesUnchecked (snark-verifier-sdk's default);
  * snark        `gen_snark_shplonk(.., Some(path))` / `read_snark(path)`      cli.rs:320,343,369,462,478-483       = bincode(Snark);
  * break points `serde_json::to_string(&agg_circuit.break_points())`         cli.rs:405-407,442-445,496-499        = Vec<Vec<usize>> as JSON;
  * SRS          `gen_srs(k)` under PARAMS_DIR                                  cli.rs:222,306                        -> ffi.ParamsKZG.read / .write.

[UPSTREAM-RECALL] Every layout below is restated from memory of halo2_proofs @ 4b42325 (/root/reference/Cargo.lock:1320-1322) and
snark-verifier-sdk @ 7011e8c (Cargo.lock:2714-2716): no file produced by the reference exists on this machine (no Rust toolchain), so
the readers are pinned only by round trips against the writers here (tests/test_formats.py).  Each layout is spelled out where it is
parsed so that a maintainer holding a real file can correct it in one place.

Encodings (RawBytes / RawBytesUnchecked): a field element is its 4 little-endian u64 Montgomery limbs — byte for byte the ABI form of
include/zkhip.h, so columns are uploaded without conversion; a G1 point is x then y (64 bytes); lengths are u32 BIG-endian.
"""
import json
import struct

import numpy as np


# ----------------------------------------------------------------------------- break points
def read_break_points(path, k=None):
    """MultiPhaseThreadBreakPoints = Vec<Vec<usize>> (/root/reference/src/helpers.rs:201 returns it as Vec<Vec<usize>>): per phase, the
    row at which each advice column's thread ends.  A phase with b break points occupies b + 1 advice columns — the aggregation
    circuit's real column count (cli.rs:493 `calculate_params(Some(10))`) can be read off this file."""
    with open(path) as f:
        bp = json.load(f)
    if not isinstance(bp, list) or any(not isinstance(p, list) or any(not isinstance(v, int) or isinstance(v, bool) or v < 0 for v in p) for p in bp):
        raise ValueError(f"{path}: not a Vec<Vec<usize>>")
    if k is not None and any(v >= (1 << k) for p in bp for v in p):
        raise ValueError(f"{path}: a break point is not below 2^{k}")
    return bp


def write_break_points(path, break_points):
    with open(path, "w") as f:
        f.write(json.dumps([[int(v) for v in p] for p in break_points], separators=(",", ":")))      # serde_json::to_string: no spaces


def advice_columns_from_break_points(break_points):
    """advice columns per phase that a halo2-lib circuit with these break points assigns (b break points -> b + 1 columns)"""
    return [len(p) + 1 for p in break_points]


# ----------------------------------------------------------------------------- SRS (ParamsKZG::write)
class ParamsFile:
    """halo2_proofs ParamsKZG<Bn256>::write (poly/kzg/commitment.rs, SerdeFormat::RawBytes) [UPSTREAM-RECALL] on the HOST — the file the
    reference keeps under PARAMS_DIR as kzg_bn254_{k}.srs (/root/reference/src/bin/cli.rs:222): k as u32 LE | g (n points) | g_lagrange
    (n points) | g2 | s_g2; a G1 point is x then y as 4 LE u64 Montgomery limbs each, a G2 point 128 bytes (x.c0, x.c1, y.c0, y.c1).
    ffi.ParamsKZG.read / .write move the same layout to and from the device; this class is the host-only view (tests, tools)."""

    def __init__(self, k, g, g_lagrange, g2_bytes):
        self.k = int(k)
        self.g = np.ascontiguousarray(g, dtype="<u8").reshape(-1, 8)
        self.g_lagrange = np.ascontiguousarray(g_lagrange, dtype="<u8").reshape(-1, 8)
        self.g2_bytes = bytes(g2_bytes)
        n = 1 << self.k
        if len(self.g) != n or len(self.g_lagrange) != n or len(self.g2_bytes) != 256:
            raise ValueError(f"ParamsFile: k = {self.k} needs {n} + {n} G1 points and 256 G2 bytes (got {len(self.g)}, {len(self.g_lagrange)}, {len(self.g2_bytes)})")

    @classmethod
    def parse(cls, buf):
        buf = bytes(buf)
        if len(buf) < 4:
            raise ValueError("params file truncated")
        k = int.from_bytes(buf[:4], "little")
        if not 1 <= k <= 28:
            raise ValueError(f"k = {k} is not a KZG parameter file")
        n = 1 << k
        if len(buf) != 4 + 2 * n * 64 + 256:
            raise ValueError(f"params file: {len(buf)} bytes for k = {k} (expected {4 + 2 * n * 64 + 256}: another layout, or truncated)")
        pts = np.frombuffer(buf, dtype="<u8", count=2 * n * 8, offset=4).reshape(2, n, 8)
        return cls(k, pts[0], pts[1], buf[4 + 2 * n * 64:])

    def to_bytes(self):
        return self.k.to_bytes(4, "little") + self.g.tobytes() + self.g_lagrange.tobytes() + self.g2_bytes


# ----------------------------------------------------------------------------- proving key (SerdeFormat::RawBytesUnchecked)
class _Reader:
    def __init__(self, buf):
        self.b, self.o = buf, 0

    def u32be(self):
        v = struct.unpack_from(">I", self.b, self.o)[0]
        self.o += 4
        return v

    def take(self, n):
        if self.o + n > len(self.b):
            raise ValueError("proving key file truncated")
        v = self.b[self.o:self.o + n]
        self.o += n
        return v

    def scalars(self, count):
        return np.frombuffer(self.take(count * 32), dtype="<u8").reshape(count, 4)

    def points(self, count):
        return np.frombuffer(self.take(count * 64), dtype="<u8").reshape(count, 8)

    def polynomial(self):
        """Polynomial::write: u32 BE length, then the values"""
        return self.scalars(self.u32be())

    def polynomial_slice(self):
        """write_polynomial_slice: u32 BE count, then the polynomials"""
        return [self.polynomial() for _ in range(self.u32be())]


class ProvingKeyFile:
    """halo2_proofs plonk::ProvingKey<G1Affine> as ProvingKey::write lays it out [UPSTREAM-RECALL]:

        vk:  k (u32 BE) | fixed commitment count (u32 BE) | fixed commitments | permutation commitments (one per permutation column,
             NO count: the reader knows it from the ConstraintSystem) | selectors: for each selector, n bools packed 8 per byte, LSB first
        l0 | l_last | l_active_row                      (Polynomial: u32 BE length + extended_n scalars each)
        fixed_values | fixed_polys | fixed_cosets        (polynomial slices: u32 BE count, then polynomials)
        permutation: permutations | polys | cosets       (polynomial slices: the sigma columns in Lagrange, coefficient and coset form)

    `read_pk::<C>` rebuilds the ConstraintSystem from the circuit (C::configure) to learn what the file does not say: the number of
    permutation columns and of selectors.  Here they are arguments.  The arrays are numpy views of the (memory-mapped) file in the
    ABI form; to_device() uploads them."""

    FIELDS = ("k", "fixed_commitments", "permutation_commitments", "selectors", "l0", "l_last", "l_active_row", "fixed_values", "fixed_polys",
              "fixed_cosets", "permutations", "permutation_polys", "permutation_cosets")

    def __init__(self, **kw):
        for f in self.FIELDS:
            setattr(self, f, kw[f])

    @classmethod
    def read(cls, path, n_perm_columns, n_selectors):
        buf = np.memmap(path, dtype=np.uint8, mode="r")
        r = _Reader(buf)
        k = r.u32be()
        if not 1 <= k <= 28:
            raise ValueError(f"{path}: k = {k}")
        n = 1 << k
        fixed_commitments = r.points(r.u32be())
        permutation_commitments = r.points(n_perm_columns)
        sel_bytes = (n + 7) // 8
        selectors = [np.unpackbits(np.frombuffer(r.take(sel_bytes), dtype=np.uint8), bitorder="little")[:n].astype(bool) for _ in range(n_selectors)]
        l0, l_last, l_active_row = r.polynomial(), r.polynomial(), r.polynomial()
        fixed_values, fixed_polys, fixed_cosets = r.polynomial_slice(), r.polynomial_slice(), r.polynomial_slice()
        permutations, permutation_polys, permutation_cosets = r.polynomial_slice(), r.polynomial_slice(), r.polynomial_slice()
        if r.o != len(buf):
            raise ValueError(f"{path}: {len(buf) - r.o} trailing bytes (wrong n_perm_columns / n_selectors, or another layout)")
        pk = cls(k=k, fixed_commitments=fixed_commitments, permutation_commitments=permutation_commitments, selectors=selectors, l0=l0,
                 l_last=l_last, l_active_row=l_active_row, fixed_values=fixed_values, fixed_polys=fixed_polys, fixed_cosets=fixed_cosets,
                 permutations=permutations, permutation_polys=permutation_polys, permutation_cosets=permutation_cosets)
        pk.check()
        return pk

    def check(self):
        n = 1 << self.k
        en = len(self.l0)
        if en < n or en & (en - 1) or len(self.l_last) != en or len(self.l_active_row) != en:
            raise ValueError("l0 / l_last / l_active_row: not three extended-domain polynomials")
        nf, npm = len(self.fixed_values), len(self.permutations)
        if len(self.fixed_polys) != nf or len(self.fixed_cosets) != nf or len(self.fixed_commitments) != nf:
            raise ValueError("fixed columns: the three forms and the commitments differ in count")
        if len(self.permutation_polys) != npm or len(self.permutation_cosets) != npm or len(self.permutation_commitments) != npm:
            raise ValueError("permutation columns: the three forms and the commitments differ in count")
        for group, size in ((self.fixed_values, n), (self.fixed_polys, n), (self.fixed_cosets, en), (self.permutations, n), (self.permutation_polys, n),
                            (self.permutation_cosets, en)):
            if any(len(p) != size for p in group):
                raise ValueError("a polynomial has the wrong length")

    @property
    def extended_k(self):
        return int(len(self.l0)).bit_length() - 1

    def write(self, path):
        n = 1 << self.k
        with open(path, "wb") as f:
            f.write(struct.pack(">I", self.k))
            f.write(struct.pack(">I", len(self.fixed_commitments)))
            f.write(np.ascontiguousarray(self.fixed_commitments, dtype="<u8").tobytes())
            f.write(np.ascontiguousarray(self.permutation_commitments, dtype="<u8").tobytes())
            for s in self.selectors:
                bits = np.zeros((n + 7) // 8 * 8, dtype=np.uint8)
                bits[:n] = np.asarray(s, dtype=np.uint8)
                f.write(np.packbits(bits, bitorder="little").tobytes())

            def poly(p):
                f.write(struct.pack(">I", len(p)))
                f.write(np.ascontiguousarray(p, dtype="<u8").tobytes())

            def pslice(ps):
                f.write(struct.pack(">I", len(ps)))
                for p in ps:
                    poly(p)

            poly(self.l0), poly(self.l_last), poly(self.l_active_row)
            pslice(self.fixed_values), pslice(self.fixed_polys), pslice(self.fixed_cosets)
            pslice(self.permutations), pslice(self.permutation_polys), pslice(self.permutation_cosets)

    def to_device(self, ctx):
        """-> dict of device tensors in the shape zk_proving_key wants: fixed_lagrange / fixed_coeff / fixed_cosets, sigma_* and the l cosets"""
        up = lambda ps: [ctx.to_device(np.ascontiguousarray(p)) for p in ps]
        return dict(fixed_lagrange=up(self.fixed_values), fixed_coeff=up(self.fixed_polys), fixed_cosets=up(self.fixed_cosets),
                    sigma_lagrange=up(self.permutations), sigma_coeff=up(self.permutation_polys), sigma_cosets=up(self.permutation_cosets),
                    l0=ctx.to_device(np.ascontiguousarray(self.l0)), l_last=ctx.to_device(np.ascontiguousarray(self.l_last)),
                    l_active=ctx.to_device(np.ascontiguousarray(self.l_active_row)))

    @classmethod
    def from_prover(cls, prover, fixed_commitments, permutation_commitments, selectors=()):
        """the synthetic keygen of prover.Prover as a ProvingKey file (host arrays)"""
        h = prover.b.to_host
        return cls(k=prover.shape.k, fixed_commitments=np.asarray(fixed_commitments, dtype=np.uint64).reshape(-1, 8),
                   permutation_commitments=np.asarray(permutation_commitments, dtype=np.uint64).reshape(-1, 8), selectors=list(selectors),
                   l0=h(prover.l0), l_last=h(prover.l_last), l_active_row=h(prover.l_active),
                   fixed_values=[h(c) for c in prover.fixed_lagrange], fixed_polys=[h(c) for c in prover.fixed_coeff],
                   fixed_cosets=[h(c) for c in prover.fixed_cosets], permutations=[h(c) for c in prover.sigma_lagrange],
                   permutation_polys=[h(c) for c in prover.sigma_coeff], permutation_cosets=[h(c) for c in prover.sigma_cosets])


# ----------------------------------------------------------------------------- Snark (bincode)
class SnarkFile:
    """snark_verifier_sdk::Snark { protocol: PlonkProtocol<G1Affine>, instances: Vec<Vec<Fr>>, proof: Vec<u8> } through bincode's default
    options (fixed-width little-endian integers, u64 lengths) [UPSTREAM-RECALL]:

        protocol  — a nested serde structure (domain, preprocessed commitments, evaluations, queries, quotient, transcript initial state,
                    linearization, accumulator indices ...).  bincode is not self-describing and the exact field list of the pinned
                    revision is not reproducible from memory, so it is carried as OPAQUE bytes: a caller gives its length (written
                    beside the file by a Rust-side helper) or it is located by the trailing-fields consistency scan of `read`;
        instances — u64 count, then per column u64 count and the elements as 32 canonical little-endian bytes each (Fr's serde form);
        proof     — u64 length, then the transcript bytes: exactly what zkhip_create_proof_ex's transcript writer emits.

    The path needs `instances` and `proof` (they become the aggregation circuit's public inputs / witness, cli.rs:478-483); the protocol
    is consumed by the in-circuit verifier, which is out of scope."""

    R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001

    def __init__(self, protocol, instances, proof):
        self.protocol, self.instances, self.proof = bytes(protocol), [[int(v) for v in col] for col in instances], bytes(proof)

    def tail_bytes(self):
        out = struct.pack("<Q", len(self.instances))
        for col in self.instances:
            out += struct.pack("<Q", len(col)) + b"".join(v.to_bytes(32, "little") for v in col)
        return out + struct.pack("<Q", len(self.proof)) + self.proof

    def write(self, path):
        with open(path, "wb") as f:
            f.write(self.protocol + self.tail_bytes())

    @classmethod
    def _parse_tail(cls, buf, off):
        """instances + proof starting at `off`; None unless they end exactly at the end of the buffer with canonical field elements"""
        try:
            (ncols,) = struct.unpack_from("<Q", buf, off)
            off += 8
            if ncols > 64:
                return None
            cols = []
            for _ in range(ncols):
                (cnt,) = struct.unpack_from("<Q", buf, off)
                off += 8
                if cnt > (1 << 24) or off + 32 * cnt > len(buf):
                    return None
                col = [int.from_bytes(buf[off + 32 * i:off + 32 * i + 32], "little") for i in range(cnt)]
                if any(v >= cls.R for v in col):
                    return None
                cols.append(col)
                off += 32 * cnt
            (plen,) = struct.unpack_from("<Q", buf, off)
            off += 8
            if off + plen != len(buf):
                return None
            return cols, bytes(buf[off:])
        except struct.error:
            return None

    @classmethod
    def read(cls, path, protocol_len=None):
        buf = open(path, "rb").read()
        if protocol_len is not None:
            got = cls._parse_tail(buf, protocol_len)
            if got is None:
                raise ValueError(f"{path}: instances / proof do not parse at offset {protocol_len}")
            return cls(buf[:protocol_len], got[0], got[1])
        # no length given: the (instances, proof) suffix is the unique offset from which both parse and end at the end of the file
        hits = [(o, t) for o in range(0, max(1, len(buf) - 15)) if (t := cls._parse_tail(buf, o)) is not None and len(t[1]) >= 64]
        # a small last instance value ends in zero bytes: the 8 of them before the proof's length read as "zero instance columns" followed by
        # that very length — an artefact of the true suffix, dropped when a candidate WITH instance columns exists
        if len(hits) > 1 and any(t[0] for _, t in hits):
            hits = [(o, t) for o, t in hits if t[0]]
        if len(hits) != 1:
            raise ValueError(f"{path}: {len(hits)} candidate offsets for the instances / proof suffix; pass protocol_len")
        o, (cols, proof) = hits[0]
        return cls(buf[:o], cols, proof)
