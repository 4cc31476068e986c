"""ctypes binding of oracle/libzkoracle.so — TEST INFRASTRUCTURE ONLY (parity unpinned, see zkoracle.h).

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Arrays are numpy uint64 with a trailing dimension of 4 (field element), 8 (affine) or 12 (Jacobian),
Montgomery form.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
U64P = C.POINTER(C.c_uint64)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libzkoracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libzkoracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.zko_splitmix64.restype = C.c_uint64
        _LIB.zko_splitmix64.argtypes = [C.c_uint64]
        _LIB.zko_domain_new.restype = C.c_void_p
        _LIB.zko_domain_new.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p]
        _LIB.zko_domain_free.argtypes = [C.c_void_p]
        _LIB.zko_domain_extended_k.argtypes = [C.c_void_p]
        _LIB.zko_domain_extended_k.restype = C.c_uint32
        _LIB.zko_domain_quotient_poly_degree.argtypes = [C.c_void_p]
        _LIB.zko_domain_quotient_poly_degree.restype = C.c_uint32
    return _LIB


def p(a):
    return a.ctypes.data_as(C.c_void_p)


def u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def new(*shape):
    return np.zeros(shape, dtype=np.uint64)


# ---- int <-> limb helpers (canonical ints on the Python side) ----
MASK = (1 << 64) - 1


def int_to_limbs(x):
    return np.array([(x >> (64 * i)) & MASK for i in range(4)], dtype=np.uint64)


def limbs_to_int(l):
    return sum(int(v) << (64 * i) for i, v in enumerate(l))


def ints_to_arr(xs):
    return np.array([[(x >> (64 * i)) & MASK for i in range(4)] for x in xs], dtype=np.uint64).reshape(len(xs), 4)


def arr_to_ints(a):
    a = np.asarray(a).reshape(-1, 4)
    return [limbs_to_int(r) for r in a]


def _unary(fn, a):
    o = new(4)
    getattr(lib(), fn)(p(u64(a)), p(o))
    return o


def _binary(fn, a, b):
    o = new(4)
    getattr(lib(), fn)(p(u64(a)), p(u64(b)), p(o))
    return o


def fr_from_int(x):
    return _unary("zko_fr_from_repr", int_to_limbs(x))


def fr_to_int(a):
    return limbs_to_int(_unary("zko_fr_to_repr", a))


def fq_from_int(x):
    return _unary("zko_fq_from_repr", int_to_limbs(x))


def fq_to_int(a):
    return limbs_to_int(_unary("zko_fq_to_repr", a))


def fr_arr_from_ints(xs):
    """canonical ints -> (n,4) Montgomery Fr"""
    out = new(len(xs), 4)
    raw = ints_to_arr(xs)
    L = lib()
    for i in range(len(xs)):
        L.zko_fr_from_repr(p(raw[i]), C.c_void_p(out.ctypes.data + 32 * i))
    return out


def fr_arr_to_ints(a):
    a = u64(a).reshape(-1, 4)
    out = new(4)
    res = []
    L = lib()
    for i in range(a.shape[0]):
        L.zko_fr_to_repr(C.c_void_p(a.ctypes.data + 32 * i), p(out))
        res.append(limbs_to_int(out))
    return res


def affine_from_ints(pts):
    """[(x, y)] canonical -> (n, 8) Montgomery"""
    out = new(len(pts), 8)
    for i, (x, y) in enumerate(pts):
        out[i, :4] = fq_from_int(x)
        out[i, 4:] = fq_from_int(y)
    return out


def affine_to_ints(a):
    a = u64(a).reshape(-1, 8)
    return [(fq_to_int(r[:4]), fq_to_int(r[4:])) for r in a]


def g1_to_affine(jac):
    o = new(8)
    lib().zko_g1_to_affine(p(u64(jac)), p(o))
    return o


def best_multiexp(coeffs, bases, threads=1):
    coeffs = u64(coeffs).reshape(-1, 4)
    bases = u64(bases).reshape(-1, 8)
    assert coeffs.shape[0] == bases.shape[0]
    o = new(12)
    lib().zko_best_multiexp(p(coeffs), p(bases), C.c_size_t(coeffs.shape[0]), C.c_int(threads), p(o))
    return o


def best_fft(a, omega, log_n, threads=1):
    a = u64(a).copy().reshape(-1, 4)
    assert a.shape[0] == 1 << log_n
    lib().zko_best_fft(p(a), p(u64(omega)), C.c_uint32(log_n), C.c_int(threads))
    return a


def root_of_unity(k):
    o = new(4)
    lib().zko_fr_root_of_unity(C.c_uint32(k), p(o))
    return o


def fr_constants():
    z, d = new(4), new(4)
    lib().zko_fr_constants(p(z), p(d))
    return z, d


def synth_raw253(seed, n, start=0):
    out = new(n, 4)
    lib().zko_synth_fill(C.c_uint64(seed), C.c_uint64(start), C.c_size_t(n), p(out))
    return out


def fixed_base_mul(scalars, threads=1):
    scalars = u64(scalars).reshape(-1, 4)
    out = new(scalars.shape[0], 8)
    lib().zko_fixed_base_mul(p(scalars), C.c_size_t(scalars.shape[0]), p(out), C.c_int(threads))
    return out


def kzg_setup_scalars(k, s):
    n = 1 << k
    mono, lag = new(n, 4), new(n, 4)
    lib().zko_kzg_setup_scalars(C.c_uint32(k), p(u64(s)), p(mono), p(lag))
    return mono, lag


def eval_polynomial(coeffs, x):
    coeffs = u64(coeffs).reshape(-1, 4)
    o = new(4)
    lib().zko_eval_polynomial(p(coeffs), C.c_size_t(coeffs.shape[0]), p(u64(x)), p(o))
    return o


def g1_mul_gen(scalar_mont):
    g = new(8)
    lib().zko_g1_generator(p(g))
    gj = new(12)
    lib().zko_g1_from_affine(p(g), p(gj))
    o = new(12)
    lib().zko_g1_mul(p(gj), p(u64(scalar_mont)), p(o))
    return g1_to_affine(o)


def g1_to_bytes(aff):
    o = (C.c_uint8 * 32)()
    lib().zko_g1_to_bytes(p(u64(aff)), o)
    return bytes(o)


def batch_invert(a):
    a = u64(a).copy().reshape(-1, 4)
    lib().zko_batch_invert(p(a), C.c_size_t(a.shape[0]))
    return a


def _ptrs(cols):
    arr = (C.c_void_p * max(1, len(cols)))(*[c.ctypes.data for c in cols])
    return arr


def permutation_products(k, values, sigmas, chunk_len, beta, gamma, bf, blinding):
    """values/sigmas: lists of (n,4) Lagrange columns; blinding: (nsets, bf, 4).  Returns the list of z columns."""
    n = 1 << k
    values = [u64(v).reshape(n, 4) for v in values]
    sigmas = [u64(v).reshape(n, 4) for v in sigmas]
    nsets = -(-len(values) // chunk_len)
    zs = [new(n, 4) for _ in range(nsets)]
    bl = u64(blinding).reshape(nsets, bf, 4)
    lib().zko_permutation_products(C.c_uint32(k), C.c_uint32(len(values)), C.c_uint32(chunk_len), _ptrs(values), _ptrs(sigmas),
                                   p(u64(beta)), p(u64(gamma)), C.c_uint32(bf), p(bl), _ptrs(zs), C.c_int(1))
    return zs


def lookup_product(k, cin, ctab, pin, ptab, beta, gamma, bf, blinding):
    n = 1 << k
    z = new(n, 4)
    lib().zko_lookup_product(C.c_uint32(k), p(u64(cin)), p(u64(ctab)), p(u64(pin)), p(u64(ptab)), p(u64(beta)), p(u64(gamma)),
                             C.c_uint32(bf), p(u64(blinding).reshape(bf, 4)), p(z))
    return z


def permute_expression_pair(k, bf, inp, tab, blind_in, blind_tab):
    n = 1 << k
    pin, ptab = new(n, 4), new(n, 4)
    rc = lib().zko_permute_expression_pair(C.c_uint32(k), C.c_uint32(bf), p(u64(inp)), p(u64(tab)), p(u64(blind_in)), p(u64(blind_tab)),
                                           p(pin), p(ptab))
    if rc != 0:
        raise ValueError("ConstraintSystemFailure: an input value is not in the table")
    return pin, ptab


def eval_polynomials(polys, x):
    polys = [u64(q).reshape(-1, 4) for q in polys]
    out = new(len(polys), 4)
    lib().zko_eval_polynomials(_ptrs(polys), C.c_size_t(len(polys)), C.c_size_t(polys[0].shape[0]), p(u64(x)), p(out))
    return out


def linear_combination(polys, coeffs, low=None):
    """sum_j coeffs[j] * polys[j] minus the low-degree polynomial `low` (shplonk prover)."""
    polys = [u64(q).reshape(-1, 4) for q in polys]
    coeffs = u64(coeffs).reshape(len(polys), 4)
    low = u64(low).reshape(-1, 4) if low is not None and len(low) else new(0, 4)
    out = new(polys[0].shape[0], 4)
    lib().zko_linear_combination(_ptrs(polys), C.c_size_t(len(polys)), C.c_size_t(polys[0].shape[0]), p(coeffs), p(low) if len(low) else None,
                                 C.c_size_t(len(low)), p(out))
    return out


def kate_division(a, roots):
    """a / prod (X - r) for r in roots (remainders dropped), same length as a."""
    a = u64(a).copy().reshape(-1, 4)
    for r in u64(roots).reshape(-1, 4):
        lib().zko_kate_division(p(a), C.c_size_t(a.shape[0]), p(r))
    return a


class Domain:
    def __init__(self, j, k, g_coset=None):
        self.h = C.c_void_p(lib().zko_domain_new(j, k, p(u64(g_coset)) if g_coset is not None else None))
        self.k = k
        self.n = 1 << k
        self.extended_k = lib().zko_domain_extended_k(self.h)
        self.extended_n = 1 << self.extended_k
        self.quotient_poly_degree = lib().zko_domain_quotient_poly_degree(self.h)
        self.omega, self.extended_omega, self.g_coset = new(4), new(4), new(4)
        lib().zko_domain_get(self.h, p(self.omega), p(self.extended_omega), p(self.g_coset))

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:   # module globals are gone at interpreter shutdown
            lib().zko_domain_free(self.h)
            self.h = None

    def lagrange_to_coeff(self, a, threads=1):
        a = u64(a).copy().reshape(self.n, 4)
        lib().zko_lagrange_to_coeff(self.h, p(a), threads)
        return a

    def coeff_to_lagrange(self, a, threads=1):
        a = u64(a).copy().reshape(self.n, 4)
        lib().zko_coeff_to_lagrange(self.h, p(a), threads)
        return a

    def coeff_to_extended(self, coeffs, threads=1):
        coeffs = u64(coeffs).reshape(-1, 4)
        out = new(self.extended_n, 4)
        lib().zko_coeff_to_extended(self.h, p(coeffs), C.c_size_t(coeffs.shape[0]), p(out), threads)
        return out

    def extended_to_coeff(self, a, threads=1):
        a = u64(a).copy().reshape(self.extended_n, 4)
        lib().zko_extended_to_coeff(self.h, p(a), threads)
        return a[: self.n * self.quotient_poly_degree]

    def extended_to_coeff_full(self, a, threads=1):
        a = u64(a).copy().reshape(self.extended_n, 4)
        lib().zko_extended_to_coeff(self.h, p(a), threads)
        return a

    def divide_by_vanishing_poly(self, a):
        a = u64(a).copy().reshape(self.extended_n, 4)
        lib().zko_divide_by_vanishing_poly(self.h, p(a))
        return a

    def l_cosets(self, blinding_factors, threads=1):
        l0, ll, la = new(self.extended_n, 4), new(self.extended_n, 4), new(self.extended_n, 4)
        lib().zko_domain_l_cosets(self.h, C.c_uint32(blinding_factors), p(l0), p(ll), p(la), threads)
        return l0, ll, la


# ---- evaluate_h marshalling (layout of zk_graph / zk_evalh_args in zkoracle.h == include/zkhip.h) ----
class ZkGraph(C.Structure):
    _fields_ = [("constants", C.c_void_p), ("rotations", C.c_void_p), ("code", C.c_void_p),
                ("n_constants", C.c_uint32), ("n_rotations", C.c_uint32), ("n_code_words", C.c_uint32),
                ("n_calculations", C.c_uint32), ("n_intermediates", C.c_uint32)]


class ZkEvalhArgs(C.Structure):
    _fields_ = [("k", C.c_uint32), ("extended_k", C.c_uint32), ("cs_degree", C.c_uint32), ("blinding_factors", C.c_uint32),
                ("extended_omega", C.c_uint64 * 4), ("g_coset", C.c_uint64 * 4), ("delta", C.c_uint64 * 4),
                ("beta", C.c_uint64 * 4), ("gamma", C.c_uint64 * 4), ("theta", C.c_uint64 * 4), ("y", C.c_uint64 * 4),
                ("n_fixed", C.c_uint32), ("n_advice", C.c_uint32), ("n_instance", C.c_uint32), ("n_challenges", C.c_uint32),
                ("fixed_cosets", C.c_void_p), ("advice_cosets", C.c_void_p), ("instance_cosets", C.c_void_p),
                ("challenges", C.c_void_p),
                ("l0", C.c_void_p), ("l_last", C.c_void_p), ("l_active_row", C.c_void_p),
                ("custom_gates", ZkGraph),
                ("n_perm_columns", C.c_uint32), ("n_perm_sets", C.c_uint32),
                ("perm_column_type", C.c_void_p), ("perm_column_index", C.c_void_p),
                ("perm_sigma_cosets", C.c_void_p), ("perm_product_cosets", C.c_void_p),
                ("n_lookups", C.c_uint32), ("_pad", C.c_uint32),
                ("lookup_graphs", C.c_void_p),
                ("lookup_product_cosets", C.c_void_p), ("lookup_input_cosets", C.c_void_p),
                ("lookup_table_cosets", C.c_void_p)]


class EvalhPack:
    """Builds a ZkEvalhArgs from Python-side descriptions; keeps every buffer alive.

    ptr_of(x) maps a column object to an address: identity for numpy host arrays (oracle) — the
    product binding passes device addresses instead."""

    def __init__(self, ptr_of=None):
        self.keep = []
        self.ptr_of = ptr_of or (lambda a: a.ctypes.data)

    def _ptr_array(self, cols):
        arr = (C.c_void_p * max(1, len(cols)))(*[self.ptr_of(c) for c in cols])
        self.keep.append(arr)
        self.keep.append(cols)
        return C.cast(arr, C.c_void_p)

    def graph(self, g, to_mont):
        consts = to_mont(g.constants)
        rots = np.array(g.rotations if g.rotations else [0], dtype=np.int32)
        code = np.array(g.code_words(), dtype=np.int32)
        self.keep += [consts, rots, code]
        return ZkGraph(consts.ctypes.data, rots.ctypes.data, code.ctypes.data, len(g.constants), len(g.rotations),
                       len(code), len(g.calculations), g.num_intermediates)

    def build(self, *, k, extended_k, cs_degree, blinding_factors, extended_omega, g_coset, delta, beta, gamma, theta, y,
              fixed, advice, instance, challenges, l0, l_last, l_active, gates_graph, perm_columns, sigma, perm_z,
              lookup_graphs, lookup_z, lookup_a, lookup_s, to_mont):
        a = ZkEvalhArgs()
        a.k, a.extended_k, a.cs_degree, a.blinding_factors = k, extended_k, cs_degree, blinding_factors
        for name, v in (("extended_omega", extended_omega), ("g_coset", g_coset), ("delta", delta), ("beta", beta),
                        ("gamma", gamma), ("theta", theta), ("y", y)):
            setattr(a, name, (C.c_uint64 * 4)(*[int(t) for t in v]))
        a.n_fixed, a.n_advice, a.n_instance = len(fixed), len(advice), len(instance)
        ch = u64(challenges).reshape(-1, 4) if len(challenges) else new(1, 4)
        self.keep.append(ch)
        a.n_challenges = len(challenges)
        a.challenges = ch.ctypes.data
        a.fixed_cosets = self._ptr_array(fixed)
        a.advice_cosets = self._ptr_array(advice)
        a.instance_cosets = self._ptr_array(instance)
        a.l0, a.l_last, a.l_active_row = self.ptr_of(l0), self.ptr_of(l_last), self.ptr_of(l_active)
        self.keep += [l0, l_last, l_active]
        a.custom_gates = self.graph(gates_graph, to_mont)
        tmap = {"advice": 0, "fixed": 1, "instance": 2}
        ptype = np.array([tmap[t] for t, _ in perm_columns] or [0], dtype=np.uint32)
        pidx = np.array([i for _, i in perm_columns] or [0], dtype=np.uint32)
        self.keep += [ptype, pidx]
        a.n_perm_columns, a.n_perm_sets = len(perm_columns), len(perm_z)
        a.perm_column_type, a.perm_column_index = ptype.ctypes.data, pidx.ctypes.data
        a.perm_sigma_cosets = self._ptr_array(sigma)
        a.perm_product_cosets = self._ptr_array(perm_z)
        a.n_lookups = len(lookup_graphs)
        garr = (ZkGraph * max(1, len(lookup_graphs)))(*[self.graph(g, to_mont) for g in lookup_graphs])
        self.keep.append(garr)
        a.lookup_graphs = C.cast(garr, C.c_void_p)
        a.lookup_product_cosets = self._ptr_array(lookup_z)
        a.lookup_input_cosets = self._ptr_array(lookup_a)
        a.lookup_table_cosets = self._ptr_array(lookup_s)
        self.args = a
        return a


def evaluate_h(pack, extended_n, threads=1):
    out = new(extended_n, 4)
    rc = lib().zko_evaluate_h(C.byref(pack.args), p(out), C.c_int(threads))
    if rc != 0:
        raise RuntimeError(f"zko_evaluate_h failed: {rc}")
    return out
