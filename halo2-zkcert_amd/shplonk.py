"""Host side of the SHPLONK multi-open prover over the device polynomial kernels.

Mirrors halo2_proofs poly/kzg/multiopen/shplonk.rs (construct_intermediate_sets) and shplonk/prover.rs
(ProverSHPLONK::create_proof) [UPSTREAM-RECALL; crate pinned at /root/reference/Cargo.lock:1320-1322; this is the
multi-open scheme behind gen_snark_shplonk, /root/reference/src/helpers.rs:233,299 and src/bin/cli.rs:320,343,369,462]:

  y, v <- transcript
  per rotation set i (commitments opened at the same set of points S_i):
      N_i(X) = sum_j y^j (P_ij(X) - R_ij(X)),  R_ij = interpolant of P_ij's evaluations on S_i
      Q_i(X) = N_i(X) / prod_{r in S_i} (X - r)
             = sum_{r in S_i} [N_i(X) / (X - r)] / prod_{s in S_i, s != r} (r - s)      (N_i vanishes on S_i: partial fractions,
               so all divisions of all sets are independent and go through one device pass)
  h(X) = sum_i v^i Q_i(X);  commit -> transcript;  u <- transcript
  L(X) = sum_i v^i Z_{T \\ S_i}(u) sum_j y^j (P_ij(X) - R_ij(u)) - Z_T(u) h(X),  T = all points
  h'(X) = L(X) / (X - u) / Z_{T \\ S_0}(u);  commit -> transcript

The O(n) work (linear combinations, divisions, the two commitments) runs on the backend; what stays here is arithmetic on
a handful of field elements (interpolation over <= 4 points, vanishing products), done with Python integers.
"""
R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001


def _inv(x):
    return pow(x % R, -1, R)


class RotationSet:
    def __init__(self, points):
        self.points = points           # ascending (BTreeSet<Fr> order = canonical integer order)
        self.commitments = []          # [(key, [eval at each point])]


def construct_intermediate_sets(queries):
    """queries: [(key, point, eval)] in query order -> (rotation sets in first-appearance order, sorted super point set)"""
    point_sets, order = {}, []
    evals = {}
    for key, point, ev in queries:
        if key not in point_sets:
            point_sets[key] = set()
            order.append(key)
        point_sets[key].add(point)
        evals[(key, point)] = ev
    sets = []
    for key in order:
        pts = point_sets[key]
        for rs in sets:
            if set(rs.points) == pts:
                break
        else:
            rs = RotationSet(sorted(pts))
            sets.append(rs)
        rs.commitments.append((key, [evals[(key, pt)] for pt in rs.points]))
    return sets, sorted({pt for _, pt, _ in queries})


def lagrange_interpolate(points, evals):
    m = len(points)
    out = [0] * m
    for i in range(m):
        num, den = [1], 1
        for j in range(m):
            if j != i:
                num = [(a - points[j] * b) % R for a, b in zip([0] + num, num + [0])]
                den = den * (points[i] - points[j]) % R
        c = evals[i] * _inv(den) % R
        for d, nd in enumerate(num):
            out[d] = (out[d] + c * nd) % R
    return out


def lagrange_basis(points):
    """coefficient lists of the m Lagrange basis polynomials over `points` (depends on the points only)"""
    m = len(points)
    basis = []
    for i in range(m):
        num, den = [1], 1
        for j in range(m):
            if j != i:
                num = [(a - points[j] * b) % R for a, b in zip([0] + num, num + [0])]
                den = den * (points[i] - points[j]) % R
        c = _inv(den)
        basis.append([nd * c % R for nd in num])
    return basis


def evaluate_vanishing_polynomial(roots, z):
    acc = 1
    for r in roots:
        acc = acc * (z - r) % R
    return acc


def _eval_small(coeffs, x):
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % R
    return acc


class ProverSHPLONK:
    """create_proof(queries, ...) over a backend offering lincomb / divide_by_linear / commit."""

    def __init__(self, backend):
        self.b = backend

    def create_proof(self, polys, queries, squeeze, write_points):
        """polys: key -> device polynomial (coefficient form, n entries).  queries: [(key, point, eval)] (ints).
        squeeze(tag) -> challenge int; write_points(tag, commitments) absorbs them (Fiat-Shamir order is upstream's:
        y, v, [h], u, [h'])."""
        b = self.b
        y = squeeze("shplonk_y")
        v = squeeze("shplonk_v")
        sets, super_points = construct_intermediate_sets(queries)
        # each set's numerator is launched as soon as its interpolants exist, so the device works while the host prepares the next
        interpolants, numerators = [], []
        for rs in sets:
            low = [0] * len(rs.points)
            coeffs, yp, r_set = [], 1, []
            basis = lagrange_basis(rs.points)
            m = len(rs.points)
            for key, evals in rs.commitments:
                r_ij = [sum(evals[i] * basis[i][d] for i in range(m)) % R for d in range(m)]
                r_set.append(r_ij)
                low = [(a + yp * c) % R for a, c in zip(low, r_ij)]
                coeffs.append(yp)
                yp = yp * y % R
            interpolants.append(r_set)
            numerators.append(b.lincomb([polys[key] for key, _ in rs.commitments], coeffs, low))
        vs = [pow(v, i, R) for i in range(len(sets))]
        src_set, roots, weights = [], [], []
        for i, rs in enumerate(sets):
            for r in rs.points:
                den = 1
                for s_ in rs.points:
                    if s_ != r:
                        den = den * (r - s_) % R
                src_set.append(i)
                roots.append(r)
                weights.append(vs[i] * _inv(den) % R)
        h_x = b.lincomb(b.divide_by_linear([numerators[i] for i in src_set], roots), weights, None)
        h1 = b.commit([h_x], lagrange=False)
        write_points("shplonk_h1", h1)
        u = squeeze("shplonk_u")
        z_diffs = [evaluate_vanishing_polynomial([p for p in super_points if p not in rs.points], u) for rs in sets]
        inv0 = _inv(z_diffs[0])
        zt = evaluate_vanishing_polynomial(super_points, u)
        cols, coeffs, const = [], [], 0
        for i, rs in enumerate(sets):
            yp = 1
            for j, (key, evals) in enumerate(rs.commitments):
                c = vs[i] * z_diffs[i] % R * yp % R * inv0 % R
                cols.append(polys[key])
                coeffs.append(c)
                const = (const + c * _eval_small(interpolants[i][j], u)) % R
                yp = yp * y % R
        cols.append(h_x)
        coeffs.append((-zt * inv0) % R)
        l_x = b.lincomb(cols, coeffs, [const])
        l_x = b.divide_by_linear([l_x], [u])[0]
        h2 = b.commit([l_x], lagrange=False)
        write_points("shplonk_h2", h2)
        return dict(y=y, v=v, u=u, h1=h1[0], h2=h2[0], h_x=h_x, l_x=l_x, rotation_sets=sets, super_point_set=super_points)
