"""First-principles big-int model of the create_proof hot path (TEST INFRASTRUCTURE ONLY).

parity unpinned: the reference (/root/reference) holds no golden vector for this path and its
arithmetic lives in un-vendored crates (halo2curves 0.4.0 @ e185711, halo2_proofs @ 4b42325,
Cargo.lock:1320-1322,1359-1361) that cannot be built here (no Rust).  This module restates the
*mathematics* (unique results: field/group elements have one canonical value) with Python ints,
independently of the C restatement in oracle/zkoracle.c, and is used only by
tests/golden/gen_golden.py to emit fixtures and by the CPU tests to cross-check the C oracle.
Nothing in the product path imports it.

Conventions follow SURVEY.md §8(a) (rows a1-a7):
  * Fr / Fq are integers mod r / p; the ABI form is Montgomery (x*2^256 mod m), 4 LE u64 limbs.
  * G1: y^2 = x^3 + 3 over Fq, generator (1, 2), affine identity encoded (0, 0).
  * best_fft: a'[j] = sum_i a[i] * omega^(i*j), natural order in and out.
  * EvaluationDomain: omega_k = ROOT_OF_UNITY^(2^(28-k)), coset shift g = ZETA (cube root of 1).
"""

P = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47  # Fq modulus
R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001  # Fr modulus
S = 28  # two-adicity of r-1
GENERATOR = 7  # multiplicative generator of Fr
T_ODD = (R - 1) >> S
ROOT_OF_UNITY = pow(GENERATOR, T_ODD, R)
DELTA = pow(GENERATOR, 1 << S, R)
# halo2curves bn256 Fr::ZETA [UPSTREAM-RECALL]: from_raw([0x8b17ea66b99c90dd, 0x5bfc41088d8daaa7,
# 0xb3c4d79d41a91758, 0]); the other primitive cube root is ZETA^2.  The domain takes the coset
# generator as a parameter, so a caller holding the other root gets consistent results.
ZETA = 0xB3C4D79D41A917585BFC41088D8DAAA78B17EA66B99C90DD
MONT_R = 1 << 256
MASK64 = (1 << 64) - 1

assert pow(ZETA, 3, R) == 1 and ZETA != 1
assert pow(ROOT_OF_UNITY, 1 << S, R) == 1 and pow(ROOT_OF_UNITY, 1 << (S - 1), R) != 1


# ----------------------------------------------------------------------------- encodings
def to_mont(x, m):
    return (x * MONT_R) % m


def from_mont(x, m):
    return (x * pow(MONT_R, -1, m)) % m


def limbs(x):
    return [(x >> (64 * i)) & MASK64 for i in range(4)]


def from_limbs(l):
    return sum(int(v) << (64 * i) for i, v in enumerate(l))


# ----------------------------------------------------------------------------- PRNG (spec)
def splitmix64(x):
    """SplitMix64 finaliser applied to a counter; the repo-wide synthetic-data generator.
    Same function in oracle/zkoracle.c (zko_splitmix64) and csrc/synth.hip."""
    x = (x + 0x9E3779B97F4A7C15) & MASK64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return z ^ (z >> 31)


def synth_word(seed, idx, limb):
    return splitmix64((seed + ((idx * 4 + limb) * 0x2545F4914F6CDD1D)) & MASK64)


def synth_raw253(seed, idx):
    """253-bit value (< r and < p) from the counter PRNG: the raw limbs of element idx."""
    l = [synth_word(seed, idx, j) for j in range(4)]
    l[3] &= 0x1FFFFFFFFFFFFFFF
    return from_limbs(l)


# ----------------------------------------------------------------------------- G1 (Jacobian ints)
INF = (0, 1, 0)


def jac_double(pt):
    X, Y, Z = pt
    if Z == 0 or Y == 0:
        return INF
    A = X * X % P
    B = Y * Y % P
    C = B * B % P
    D = 2 * ((X + B) * (X + B) - A - C) % P
    E = 3 * A % P
    F = E * E % P
    X3 = (F - 2 * D) % P
    Y3 = (E * (D - X3) - 8 * C) % P
    Z3 = 2 * Y * Z % P
    return (X3, Y3, Z3)


def jac_add(p1, p2):
    X1, Y1, Z1 = p1
    X2, Y2, Z2 = p2
    if Z1 == 0:
        return p2
    if Z2 == 0:
        return p1
    Z1Z1 = Z1 * Z1 % P
    Z2Z2 = Z2 * Z2 % P
    U1 = X1 * Z2Z2 % P
    U2 = X2 * Z1Z1 % P
    S1 = Y1 * Z2 * Z2Z2 % P
    S2 = Y2 * Z1 * Z1Z1 % P
    if U1 == U2:
        if S1 == S2:
            return jac_double(p1)
        return INF
    H = (U2 - U1) % P
    Rr = (S2 - S1) % P
    H2 = H * H % P
    H3 = H * H2 % P
    V = U1 * H2 % P
    X3 = (Rr * Rr - H3 - 2 * V) % P
    Y3 = (Rr * (V - X3) - S1 * H3) % P
    Z3 = Z1 * Z2 * H % P
    return (X3, Y3, Z3)


def jac_neg(pt):
    return (pt[0], (-pt[1]) % P, pt[2])


def to_affine(pt):
    """(x, y) with identity = (0, 0) (halo2curves G1Affine convention)."""
    X, Y, Z = pt
    if Z == 0:
        return (0, 0)
    zi = pow(Z, -1, P)
    zi2 = zi * zi % P
    return (X * zi2 % P, Y * zi2 * zi % P)


def from_affine(a):
    if a == (0, 0):
        return INF
    return (a[0], a[1], 1)


def on_curve(a):
    return a == (0, 0) or (a[1] * a[1] - a[0] ** 3 - 3) % P == 0


G1_GEN = (1, 2, 1)


def scalar_mul(k, pt):
    k %= R
    acc = INF
    for bit in bin(k)[2:] if k else "":
        acc = jac_double(acc)
        if bit == "1":
            acc = jac_add(acc, pt)
    return acc


def msm_naive(scalars, points_affine):
    """sum_i scalars[i] * points[i] by independent double-and-add (no bucket method)."""
    acc = INF
    for s, a in zip(scalars, points_affine):
        acc = jac_add(acc, scalar_mul(s, from_affine(a)))
    return to_affine(acc)


def compress(a):
    """32-byte LE x with the y-sign in bit 6 of byte 31 and identity in bit 7
    [UPSTREAM-RECALL halo2curves 0.4.0 new_curve_impl, SURVEY §8(c) open item 1]."""
    if a == (0, 0):
        b = bytearray(32)
        b[31] |= 0x80
        return bytes(b)
    b = bytearray(a[0].to_bytes(32, "little"))
    b[31] |= (a[1] & 1) << 6
    return bytes(b)


# ----------------------------------------------------------------------------- NTT
def omega_for(k):
    return pow(ROOT_OF_UNITY, 1 << (S - k), R)


def dft_naive(a, omega):
    n = len(a)
    pw = [pow(omega, i, R) for i in range(n)]
    return [sum(a[i] * pw[(i * j) % n] for i in range(n)) % R for j in range(n)]


def fft(a, omega):
    """Recursive radix-2; validated against dft_naive in the tests."""
    n = len(a)
    if n == 1:
        return list(a)
    w2 = omega * omega % R
    ev = fft(a[0::2], w2)
    od = fft(a[1::2], w2)
    out = [0] * n
    w = 1
    h = n // 2
    for j in range(h):
        t = w * od[j] % R
        out[j] = (ev[j] + t) % R
        out[j + h] = (ev[j] - t) % R
        w = w * omega % R
    return out


def ifft(a, omega):
    n = len(a)
    ninv = pow(n, -1, R)
    return [x * ninv % R for x in fft(a, pow(omega, -1, R))]


# ----------------------------------------------------------------------------- EvaluationDomain
class Domain:
    """halo2_proofs::poly::EvaluationDomain::new(j, k) [UPSTREAM-RECALL poly/domain.rs]."""

    def __init__(self, j, k, zeta=ZETA):
        self.k = k
        self.n = 1 << k
        self.quotient_poly_degree = j - 1
        ek = k
        while (1 << ek) < self.n * self.quotient_poly_degree:
            ek += 1
        self.extended_k = ek
        self.extended_n = 1 << ek
        self.extended_omega = omega_for(ek)
        self.omega = omega_for(k)
        self.g_coset = zeta
        self.g_coset_inv = zeta * zeta % R
        # t_evaluations[i] = 1 / ((g * w_ext^i)^n - 1), period 2^(ek-k)
        self.t_evaluations = []
        step = pow(self.extended_omega, self.n, R)
        cur = pow(self.g_coset, self.n, R)
        for _ in range(1 << (ek - k)):
            self.t_evaluations.append(pow((cur - 1) % R, -1, R))
            cur = cur * step % R

    def lagrange_to_coeff(self, a):
        return ifft(a, self.omega)

    def coeff_to_lagrange(self, a):
        return fft(a, self.omega)

    def coeff_to_extended(self, a):
        a = list(a) + [0] * (self.extended_n - len(a))
        cp = [1, self.g_coset, self.g_coset_inv]
        a = [x * cp[i % 3] % R for i, x in enumerate(a)]
        return fft(a, self.extended_omega)

    def extended_to_coeff(self, a):
        a = ifft(a, self.extended_omega)
        cp = [1, self.g_coset_inv, self.g_coset]
        a = [x * cp[i % 3] % R for i, x in enumerate(a)]
        return a[: self.n * self.quotient_poly_degree]

    def divide_by_vanishing_poly(self, a):
        m = len(self.t_evaluations)
        return [x * self.t_evaluations[i % m] % R for i, x in enumerate(a)]

    def coset_point(self, i):
        return self.g_coset * pow(self.extended_omega, i, R) % R


def poly_eval(coeffs, x):
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % R
    return acc


# ----------------------------------------------------------------------------- keygen cosets
def l_cosets(dom, blinding_factors):
    """l_0, l_last, l_active_row on the extended coset [UPSTREAM-RECALL plonk/keygen.rs]."""
    n = dom.n

    def ext(lagr):
        return dom.coeff_to_extended(dom.lagrange_to_coeff(lagr))

    l0 = ext([1] + [0] * (n - 1))
    lb = [0] * n
    for i in range(blinding_factors):
        lb[n - 1 - i] = 1
    l_blind = ext(lb)
    ll = [0] * n
    ll[n - blinding_factors - 1] = 1
    l_last = ext(ll)
    l_active = [(1 - (a + b)) % R for a, b in zip(l_last, l_blind)]
    return l0, l_last, l_active


# ----------------------------------------------------------------------------- evaluate_h (direct)
def eval_expr(e, idx, rot_scale, isize, fixed, advice, instance, challenges):
    k = e[0]
    if k == "const":
        return e[1] % R
    if k in ("fixed", "advice", "instance"):
        col = {"fixed": fixed, "advice": advice, "instance": instance}[k][e[1]]
        return col[(idx + e[2] * rot_scale) % isize]
    if k == "challenge":
        return challenges[e[1]]
    if k == "neg":
        return (-eval_expr(e[1], idx, rot_scale, isize, fixed, advice, instance, challenges)) % R
    if k == "sum":
        return (eval_expr(e[1], idx, rot_scale, isize, fixed, advice, instance, challenges)
                + eval_expr(e[2], idx, rot_scale, isize, fixed, advice, instance, challenges)) % R
    if k == "prod":
        return (eval_expr(e[1], idx, rot_scale, isize, fixed, advice, instance, challenges)
                * eval_expr(e[2], idx, rot_scale, isize, fixed, advice, instance, challenges)) % R
    if k == "scaled":
        return eval_expr(e[1], idx, rot_scale, isize, fixed, advice, instance, challenges) * e[2] % R
    raise ValueError(k)


def evaluate_h_direct(dom, cs, cosets, ch):
    """Quotient numerator on the extended coset from the constraint definitions themselves
    (expression trees, not the flattened graph) [UPSTREAM-RECALL plonk/evaluation.rs evaluate_h:
    term order = gates, permutation (l0, l_last, links, products), lookups (5 terms each)].

    cs: dict(gates=[expr], lookups=[(inputs, tables)], perm_columns=[(type, idx)], degree, blinding_factors)
    cosets: dict(fixed, advice, instance, l0, l_last, l_active, sigma, perm_z, lookup_z, lookup_a, lookup_s)
    ch: dict(beta, gamma, theta, y, challenges)
    """
    isize = dom.extended_n
    rs = 1 << (dom.extended_k - dom.k)
    fx, ad, ins = cosets["fixed"], cosets["advice"], cosets["instance"]
    l0, l_last, l_act = cosets["l0"], cosets["l_last"], cosets["l_active"]
    beta, gamma, theta, y = ch["beta"], ch["gamma"], ch["theta"], ch["y"]
    bf = cs["blinding_factors"]
    chunk = cs["degree"] - 2
    out = []
    for idx in range(isize):
        terms = [eval_expr(g, idx, rs, isize, fx, ad, ins, ch["challenges"]) for g in cs["gates"]]
        zs = cosets["perm_z"]
        if zs:
            r_next = (idx + rs) % isize
            r_last = (idx - (bf + 1) * rs) % isize
            terms.append((1 - zs[0][idx]) * l0[idx] % R)
            terms.append((zs[-1][idx] ** 2 - zs[-1][idx]) * l_last[idx] % R)
            for s in range(1, len(zs)):
                terms.append((zs[s][idx] - zs[s - 1][r_last]) * l0[idx] % R)
            x = dom.coset_point(idx)
            cur = beta * x % R  # beta * delta^j * X, j over permutation columns
            for s in range(len(zs)):
                cols = cs["perm_columns"][s * chunk:(s + 1) * chunk]
                sig = cosets["sigma"][s * chunk:(s + 1) * chunk]
                left, right = zs[s][r_next], zs[s][idx]
                for (typ, ci), sg in zip(cols, sig):
                    v = {"advice": ad, "fixed": fx, "instance": ins}[typ][ci][idx]
                    left = left * (v + beta * sg[idx] + gamma) % R
                    right = right * (v + cur + gamma) % R
                    cur = cur * DELTA % R
                terms.append((left - right) * l_act[idx] % R)
        for li, (inputs, tables) in enumerate(cs["lookups"]):
            z, a, s_ = cosets["lookup_z"][li], cosets["lookup_a"][li], cosets["lookup_s"][li]
            r_next = (idx + rs) % isize
            r_prev = (idx - rs) % isize

            def compress_(exprs):
                acc = 0
                for ex in exprs:
                    acc = (acc * theta + eval_expr(ex, idx, rs, isize, fx, ad, ins, ch["challenges"])) % R
                return acc

            table_value = (compress_(inputs) + beta) * (compress_(tables) + gamma) % R
            terms.append((1 - z[idx]) * l0[idx] % R)
            terms.append((z[idx] ** 2 - z[idx]) * l_last[idx] % R)
            terms.append((z[r_next] * (a[idx] + beta) * (s_[idx] + gamma) - z[idx] * table_value) * l_act[idx] % R)
            terms.append((a[idx] - s_[idx]) * l0[idx] % R)
            terms.append((a[idx] - s_[idx]) * (a[idx] - a[r_prev]) * l_act[idx] % R)
        acc = 0
        for t in terms:
            acc = (acc * y + t) % R
        out.append(acc)
    return out


# ----------------------------------------------------------------------------- grand products (a8)
def permutation_products(k, values, sigmas, chunk_len, beta, gamma, bf, blinding):
    """z polynomials of the permutation argument, Lagrange form, straight from the definition:
    z_s[0] = last z of the previous set (1 for the first); z_s[i+1] = z_s[i] * prod_j (v_j + delta^j w^i beta + gamma)
    / (v_j + beta sigma_j + gamma); the last bf rows are the given blinding values."""
    n = 1 << k
    w = omega_for(k)
    out = []
    last = 1
    j_global = 0
    for s in range(0, len(values), chunk_len):
        cols = list(range(s, min(s + chunk_len, len(values))))
        z = [last]
        for i in range(n - 1):
            num = den = 1
            for jj, c in enumerate(cols):
                num = num * (values[c][i] + pow(DELTA, j_global + jj, R) * pow(w, i, R) * beta + gamma) % R
                den = den * (values[c][i] + beta * sigmas[c][i] + gamma) % R
            z.append(z[-1] * num * pow(den, -1, R) % R)
        j_global += len(cols)
        for t in range(bf):
            z[n - bf + t] = blinding[len(out)][t]
        last = z[n - bf - 1]
        out.append(z)
    return out


def lookup_product(k, cin, ctab, pin, ptab, beta, gamma, bf, blinding):
    n = 1 << k
    z = [1]
    for i in range(n - bf - 1):
        num = (cin[i] + beta) * (ctab[i] + gamma) % R
        den = (pin[i] + beta) * (ptab[i] + gamma) % R
        z.append(z[-1] * num * pow(den, -1, R) % R)
    return z + list(blinding)


def permute_expression_pair(k, bf, inp, tab, blind_in, blind_tab):
    """lookup permutation from its definition: A' = sorted input over the usable rows; S' = a permutation of the table
    with S'[i] = A'[i] wherever A'[i] != A'[i-1]; remaining table values ascending into the remaining rows descending."""
    from collections import Counter

    n = 1 << k
    u = n - (bf + 1)
    a = sorted(inp[:u])
    left = Counter(tab[:u])
    s = [None] * u
    rep = []
    for i, v in enumerate(a):
        if i == 0 or v != a[i - 1]:
            s[i] = v
            if left[v] <= 0:
                raise ValueError("ConstraintSystemFailure")
            left[v] -= 1
        else:
            rep.append(i)
    for v in sorted(left):
        for _ in range(left[v]):
            s[rep.pop()] = v
    assert not rep
    return a + list(blind_in), s + list(blind_tab)


# ----------------------------------------------------------------------------- SHPLONK multi-open
# halo2_proofs src/poly/kzg/multiopen/shplonk.rs (construct_intermediate_sets), shplonk/prover.rs (create_proof),
# shplonk/verifier.rs (verify_proof) [UPSTREAM-RECALL; crate pinned at /root/reference/Cargo.lock:1320-1322; reached through
# gen_snark_shplonk, /root/reference/src/helpers.rs:233,299].  Polynomials are coefficient lists of ints mod R.
def construct_intermediate_sets(queries):
    """queries: [(commitment_id, point, eval)] in query order.  Commitments are grouped by the SET of points they are opened at;
    sets appear in the order their first commitment does, points inside a set ascending (BTreeSet over Fr's Ord = canonical
    integer order).  Returns (rotation_sets, super_point_set): rotation_sets = [dict(points=[..], commitments=[(id, evals)])]."""
    super_points = sorted({pt for _, pt, _ in queries})
    per_commitment = []            # [(id, set(points))], first-appearance order
    for cid, pt, _ in queries:
        for ent in per_commitment:
            if ent[0] == cid:
                ent[1].add(pt)
                break
        else:
            per_commitment.append((cid, {pt}))
    sets = []                      # [(frozenset(points), [ids])]
    for cid, pts in per_commitment:
        for ent in sets:
            if ent[0] == pts:
                ent[1].append(cid)
                break
        else:
            sets.append((set(pts), [cid]))
    ev = {(cid, pt): e for cid, pt, e in queries}
    rotation_sets = []
    for pts, ids in sets:
        points = sorted(pts)
        rotation_sets.append(dict(points=points, commitments=[(cid, [ev[(cid, pt)] for pt in points]) for cid in ids]))
    return rotation_sets, super_points


def lagrange_interpolate(points, evals):
    """coefficients (len(points)) of the polynomial through (points[i], evals[i])"""
    m = len(points)
    out = [0] * m
    for i in range(m):
        num = [1]                  # prod_{j != i} (X - x_j)
        den = 1
        for j in range(m):
            if j == i:
                continue
            num = [(a - points[j] * b) % R for a, b in zip([0] + num, num + [0])]
            den = den * (points[i] - points[j]) % R
        c = evals[i] * pow(den, R - 2, R) % R
        for d in range(len(num)):
            out[d] = (out[d] + c * num[d]) % R
    return out


def eval_vanishing(roots, z):
    acc = 1
    for r in roots:
        acc = acc * (z - r) % R
    return acc


def kate_division(a, root):
    """a / (X - root), remainder dropped: len(a) - 1 coefficients (arithmetic::kate_division)"""
    q = [0] * (len(a) - 1)
    s = 0
    for j in range(len(a) - 1, 0, -1):
        s = (a[j] + root * s) % R
        q[j - 1] = s
    return q


def shplonk_quotient(polys, rotation_sets, y, v, n):
    """h(X) = sum_i v^i [sum_j y^j (P_ij - R_ij)](X) / Z_i(X), R_ij = the interpolant of P_ij's evals on set i's points"""
    h = [0] * n
    vp = 1
    for rs in rotation_sets:
        num = [0] * n
        yp = 1
        for cid, evals in rs["commitments"]:
            low = lagrange_interpolate(rs["points"], evals)
            pj = polys[cid]
            for d in range(n):
                num[d] = (num[d] + yp * (pj[d] - (low[d] if d < len(low) else 0))) % R
            yp = yp * y % R
        for r in rs["points"]:
            num = kate_division(num, r)
        num += [0] * (n - len(num))
        for d in range(n):
            h[d] = (h[d] + vp * num[d]) % R
        vp = vp * v % R
    return h


def shplonk_linearisation(polys, rotation_sets, super_points, y, v, u, h, n):
    """[sum_i v^i Z_{T\\S_i}(u) sum_j y^j (P_ij(X) - R_ij(u)) - Z_T(u) h(X)] / (X - u), scaled by 1 / Z_{T\\S_0}(u)"""
    l = [0] * n
    vp = 1
    z_diffs = []
    for rs in rotation_sets:
        z_i = eval_vanishing([p_ for p_ in super_points if p_ not in rs["points"]], u)
        z_diffs.append(z_i)
        yp = 1
        for cid, evals in rs["commitments"]:
            r_u = poly_eval(lagrange_interpolate(rs["points"], evals), u)
            c = vp * z_i % R * yp % R
            pj = polys[cid]
            for d in range(n):
                l[d] = (l[d] + c * pj[d]) % R
            l[0] = (l[0] - c * r_u) % R
            yp = yp * y % R
        vp = vp * v % R
    zt = eval_vanishing(super_points, u)
    for d in range(n):
        l[d] = (l[d] - zt * h[d]) % R
    assert poly_eval(l, u) == 0, "shplonk: linearisation does not vanish at u"
    q = kate_division(l, u) + [0]
    inv0 = pow(z_diffs[0], R - 2, R)
    return [c * inv0 % R for c in q]


def shplonk_verify(commitments, rotation_sets, super_points, y, v, u, h1, h2, s):
    """The verifier's pairing check e(h2, [s]_2) = e(L, [1]_2) restated with the trapdoor s: s * h2 == L in G1, where
    L = sum_i v^i z_i (sum_j y^j C_ij) - [sum_i v^i z_i sum_j y^j R_ij(u)] G - z_0 h1 + u h2, z_i = Z_{T\\S_i}(u) / Z_{T\\S_0}(u),
    z_0 = Z_T(u) / Z_{T\\S_0}(u).  commitments: id -> affine point; h1, h2 affine."""
    G = (1, 2)
    inv0 = None
    acc = INF
    r_acc = 0
    vp = 1
    for i, rs in enumerate(rotation_sets):
        z_i = eval_vanishing([p_ for p_ in super_points if p_ not in rs["points"]], u)
        if i == 0:
            inv0 = pow(z_i, R - 2, R)
        z_i = z_i * inv0 % R
        yp = 1
        for cid, evals in rs["commitments"]:
            c = vp * z_i % R * yp % R
            acc = jac_add(acc, scalar_mul(c, from_affine(commitments[cid])))
            r_acc = (r_acc + c * poly_eval(lagrange_interpolate(rs["points"], evals), u)) % R
            yp = yp * y % R
        vp = vp * v % R
    z_0 = eval_vanishing(super_points, u) * inv0 % R
    acc = jac_add(acc, scalar_mul((-r_acc) % R, from_affine(G)))
    acc = jac_add(acc, scalar_mul((-z_0) % R, from_affine(h1)))
    acc = jac_add(acc, scalar_mul(u, from_affine(h2)))
    return to_affine(acc) == to_affine(scalar_mul(s % R, from_affine(h2)))


# ----------------------------------------------------------------------------- PLONK verifier (algebraic)
# halo2_proofs src/plonk/verifier.rs verify_proof, permutation/verifier.rs, lookup/verifier.rs, vanishing/verifier.rs
# [UPSTREAM-RECALL], restated on integers.  The final pairing is replaced by the same equation under the SRS trapdoor
# (shplonk_verify).  Fiat-Shamir is NOT re-derived here: the challenges are inputs (the schedule uses a stand-in hash).
def eval_expr_at(e, get):
    """expression tree -> value, get(kind, column, rotation) supplying the opened evaluations"""
    k = e[0]
    if k == "const":
        return e[1] % R
    if k in ("fixed", "advice", "instance"):
        return get(k, e[1], e[2])
    if k == "neg":
        return (-eval_expr_at(e[1], get)) % R
    if k == "sum":
        return (eval_expr_at(e[1], get) + eval_expr_at(e[2], get)) % R
    if k == "prod":
        return eval_expr_at(e[1], get) * eval_expr_at(e[2], get) % R
    if k == "scaled":
        return eval_expr_at(e[1], get) * e[2] % R
    raise ValueError(k)


def lagrange_basis_at(k, rows, x):
    """l_i(x) for i in rows: (x^n - 1) / n * w^i / (x - w^i)"""
    n = 1 << k
    w = omega_for(k)
    c = (pow(x, n, R) - 1) * pow(n, R - 2, R) % R
    return [c * pow(w, i % n, R) % R * pow((x - pow(w, i % n, R)) % R, R - 2, R) % R for i in rows]


def plonk_expected_h(vk, instance_cols, evals, ch):
    """The verifier's value of the quotient at x from the opened evaluations.
    vk: dict(k, degree, blinding_factors, gates, lookups, perm_columns); instance_cols: full Lagrange columns (ints);
    evals: ((kind, index), rotation) -> int with kinds advice, fixed, sigma, perm_z, lookup_z, lookup_a, lookup_s;
    ch: theta, beta, gamma, y, x."""
    k, bf = vk["k"], vk["blinding_factors"]
    n = 1 << k
    x, y, beta, gamma, theta = ch["x"], ch["y"], ch["beta"], ch["gamma"], ch["theta"]
    w = omega_for(k)
    inst_cache = {}

    def get(kind, col, rot):
        if kind == "instance":      # QUERY_INSTANCE = false for KZG: the verifier evaluates the instance polynomial itself
            if (col, rot) not in inst_cache:
                pt = x * pow(w, rot % n, R) % R
                ls = lagrange_basis_at(k, range(n), pt)
                inst_cache[(col, rot)] = sum(a * b for a, b in zip(instance_cols[col], ls)) % R
            return inst_cache[(col, rot)]
        return evals[((kind, col), rot)]

    ls = lagrange_basis_at(k, range(-(bf + 1), 1), x)       # rows n-bf-1 .. n-1, 0
    l_last, l_blind, l_0 = ls[0], sum(ls[1:bf + 1]) % R, ls[bf + 1]
    l_active = (1 - l_last - l_blind) % R
    last_rot = -(bf + 1)
    terms = [eval_expr_at(g, get) for g in vk["gates"]]
    chunk = vk["degree"] - 2
    cols = vk["perm_columns"]
    nsets = -(-len(cols) // chunk) if cols else 0
    Z = lambda s, rot: evals[(("perm_z", s), rot)]
    if nsets:
        terms.append(l_0 * (1 - Z(0, 0)) % R)
        terms.append(l_last * (Z(nsets - 1, 0) ** 2 - Z(nsets - 1, 0)) % R)
        for s in range(1, nsets):
            terms.append(l_0 * (Z(s, 0) - Z(s - 1, last_rot)) % R)
        cur = beta * x % R
        for s in range(nsets):
            left, right = Z(s, 1), Z(s, 0)
            for j in range(s * chunk, min(len(cols), (s + 1) * chunk)):
                v = get(cols[j][0], cols[j][1], 0)
                left = left * (v + beta * evals[(("sigma", j), 0)] + gamma) % R
                right = right * (v + cur + gamma) % R
                cur = cur * DELTA % R
            terms.append(l_active * (left - right) % R)
    for li, (inputs, tables) in enumerate(vk["lookups"]):
        z0, z1 = evals[(("lookup_z", li), 0)], evals[(("lookup_z", li), 1)]
        a0, am1, s0 = evals[(("lookup_a", li), 0)], evals[(("lookup_a", li), -1)], evals[(("lookup_s", li), 0)]

        def compress(exprs):
            acc = 0
            for ex in exprs:
                acc = (acc * theta + eval_expr_at(ex, get)) % R
            return acc

        terms.append(l_0 * (1 - z0) % R)
        terms.append(l_last * (z0 * z0 - z0) % R)
        terms.append(l_active * (z1 * (a0 + beta) % R * (s0 + gamma) - z0 * (compress(inputs) + beta) % R * (compress(tables) + gamma)) % R)
        terms.append(l_0 * (a0 - s0) % R)
        terms.append(l_active * (a0 - s0) % R * (a0 - am1) % R)
    acc = 0
    for t in terms:
        acc = (acc * y + t) % R
    return acc * pow((pow(x, n, R) - 1) % R, R - 2, R) % R


def plonk_verify(vk, instance_cols, commitments, h_pieces, evals, query_list, ch, h1, h2, s):
    """commitments: key -> affine point for every queried polynomial except h; h_pieces: the quotient's piece commitments;
    query_list: [(key, rotation)] in the prover's query order (h included); True iff the multi-open verifies with h's
    evaluation REPLACED by the value the constraint system dictates."""
    k = vk["k"]
    n = 1 << k
    x = ch["x"]
    w = omega_for(k)
    expected_h = plonk_expected_h(vk, instance_cols, evals, ch)
    xn = pow(x, n, R)
    hc = INF
    for piece in reversed(h_pieces):
        hc = jac_add(scalar_mul(xn, hc), from_affine(piece))
    coms = dict(commitments)
    coms[("h", 0)] = to_affine(hc)
    queries = []
    for key, rot in query_list:
        ev = expected_h if key == ("h", 0) else evals[(key, rot)]
        queries.append((key, x * pow(w, rot % n, R) % R, ev))
    rs, sp = construct_intermediate_sets(queries)
    return shplonk_verify(coms, rs, sp, ch["shplonk_y"], ch["shplonk_v"], ch["shplonk_u"], h1, h2, s)
