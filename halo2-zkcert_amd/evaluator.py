"""Keygen-side mirror of halo2_proofs::plonk::evaluation::{GraphEvaluator, Evaluator::new}
[UPSTREAM-RECALL src/plonk/evaluation.rs; crate pinned at /root/reference/Cargo.lock:1320-1322].

Flattens gate / lookup Expression trees into the (constants, rotations, calculations) form that
`evaluate_h` interprets, with the same de-duplication and operand ordering rules as upstream, and
packs it into the int32 code stream documented in include/zkhip.h.  Host logic only (runs once per
circuit); the per-row interpretation is the HIP sweep kernel.

Expressions are tuples:
  ("const", int) ("fixed", col, rot) ("advice", col, rot) ("instance", col, rot) ("challenge", i)
  ("neg", e) ("sum", a, b) ("prod", a, b) ("scaled", e, int)
"""
from dataclasses import dataclass, field

R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001

VS_CONSTANT, VS_INTERMEDIATE, VS_FIXED, VS_ADVICE, VS_INSTANCE, VS_CHALLENGE = 0, 1, 2, 3, 4, 5
VS_BETA, VS_GAMMA, VS_THETA, VS_Y, VS_PREVIOUS = 6, 7, 8, 9, 10
OP_ADD, OP_SUB, OP_MUL, OP_SQUARE, OP_DOUBLE, OP_NEGATE, OP_HORNER, OP_STORE = range(8)


@dataclass
class GraphEvaluator:
    constants: list = field(default_factory=lambda: [0, 1, 2])
    rotations: list = field(default_factory=list)
    calculations: list = field(default_factory=list)  # (op, sources tuple, target)
    num_intermediates: int = 0

    def add_rotation(self, rot):
        if rot in self.rotations:
            return self.rotations.index(rot)
        self.rotations.append(rot)
        return len(self.rotations) - 1

    def add_constant(self, c):
        c %= R
        if c in self.constants:
            return (VS_CONSTANT, self.constants.index(c), 0)
        self.constants.append(c)
        return (VS_CONSTANT, len(self.constants) - 1, 0)

    def add_calculation(self, op, sources):
        sources = tuple(sources)
        for (o, s, t) in self.calculations:
            if o == op and s == sources:
                return (VS_INTERMEDIATE, t, 0)
        target = self.num_intermediates
        self.calculations.append((op, sources, target))
        self.num_intermediates += 1
        return (VS_INTERMEDIATE, target, 0)

    def add_expression(self, e):
        ZERO, ONE, TWO = (VS_CONSTANT, 0, 0), (VS_CONSTANT, 1, 0), (VS_CONSTANT, 2, 0)
        kind = e[0]
        if kind == "const":
            return self.add_constant(e[1])
        if kind in ("fixed", "advice", "instance"):
            rot_idx = self.add_rotation(e[2])
            vs = {"fixed": VS_FIXED, "advice": VS_ADVICE, "instance": VS_INSTANCE}[kind]
            return self.add_calculation(OP_STORE, [(vs, e[1], rot_idx)])
        if kind == "challenge":
            return self.add_calculation(OP_STORE, [(VS_CHALLENGE, e[1], 0)])
        if kind == "neg":
            if e[1][0] == "const":
                return self.add_constant(-e[1][1])
            ra = self.add_expression(e[1])
            return ra if ra == ZERO else self.add_calculation(OP_NEGATE, [ra])
        if kind == "sum":
            a, b = e[1], e[2]
            if b[0] == "neg":
                ra = self.add_expression(a)
                rb = self.add_expression(b[1])
                if ra == ZERO:
                    return self.add_calculation(OP_NEGATE, [rb])
                if rb == ZERO:
                    return ra
                return self.add_calculation(OP_SUB, [ra, rb])
            ra = self.add_expression(a)
            rb = self.add_expression(b)
            if ra == ZERO:
                return rb
            if rb == ZERO:
                return ra
            return self.add_calculation(OP_ADD, [ra, rb] if ra <= rb else [rb, ra])
        if kind == "prod":
            ra = self.add_expression(e[1])
            rb = self.add_expression(e[2])
            if ra == ZERO or rb == ZERO:
                return ZERO
            if ra == ONE:
                return rb
            if rb == ONE:
                return ra
            if ra == TWO:
                return self.add_calculation(OP_DOUBLE, [rb])
            if rb == TWO:
                return self.add_calculation(OP_DOUBLE, [ra])
            if ra == rb:
                return self.add_calculation(OP_SQUARE, [ra])
            return self.add_calculation(OP_MUL, [ra, rb] if ra <= rb else [rb, ra])
        if kind == "scaled":
            f = e[2] % R
            if f == 0:
                return ZERO
            if f == 1:
                return self.add_expression(e[1])
            cst = self.add_constant(f)
            ra = self.add_expression(e[1])
            return self.add_calculation(OP_MUL, [ra, cst])
        raise ValueError(f"unknown expression {kind}")

    def code_words(self):
        """int32 stream: per calculation {op, target, nsrc, nsrc x (kind, a, b)}."""
        out = []
        for (op, src, target) in self.calculations:
            out += [op, target, len(src)]
            for s in src:
                out += list(s)
        return out


def build_custom_gates(gate_polys):
    """Evaluator::new, custom-gate part: all polys then one Horner in y over PreviousValue."""
    g = GraphEvaluator()
    parts = [g.add_expression(p) for p in gate_polys]
    g.add_calculation(OP_HORNER, [(VS_PREVIOUS, 0, 0), (VS_Y, 0, 0)] + parts)
    return g


def build_lookup(input_exprs, table_exprs):
    """Evaluator::new, per-lookup graph: (theta-compressed input + beta) * (compressed table + gamma)."""
    g = GraphEvaluator()

    def evaluate_lc(exprs):
        parts = [g.add_expression(x) for x in exprs]
        return g.add_calculation(OP_HORNER, [(VS_CONSTANT, 0, 0), (VS_THETA, 0, 0)] + parts)

    ci = evaluate_lc(input_exprs)
    ct = evaluate_lc(table_exprs)
    right_gamma = g.add_calculation(OP_ADD, [ct, (VS_GAMMA, 0, 0)])
    lc = g.add_calculation(OP_ADD, [ci, (VS_BETA, 0, 0)])
    g.add_calculation(OP_MUL, [lc, right_gamma])
    return g
