"""Backend for halo2_zkcert_amd.prover.Prover that computes with the CPU oracle (TEST INFRASTRUCTURE:
the checker for the GPU schedule and bench.py's cpu_baseline leg — never the product path)."""
import contextlib

import numpy as np

import zkoracle_py as zo


class OracleBackend:
    def __init__(self, threads=1):
        self.threads = threads

    def setup(self, k, degree, s_int):
        s = zo.fr_from_int(s_int)
        mono, lag = zo.kzg_setup_scalars(k, s)
        self.g = zo.fixed_base_mul(mono, self.threads)
        self.g_lagrange = zo.fixed_base_mul(lag, self.threads)
        self.domain = zo.Domain(degree, k)
        return self.domain

    @contextlib.contextmanager
    def overlap(self):
        yield

    def join(self):
        pass

    def fr(self, x):
        return zo.fr_from_int(x)

    def fr_many(self, xs):
        return zo.fr_arr_from_ints(list(xs)) if len(xs) else np.zeros((0, 4), dtype=np.uint64)

    def synth(self, n, seed):
        return zo.synth_raw253(seed, n)

    def synth_small(self, n, seed, bits_per_mille, word_bits):
        """include/zkhip.h zkhip_synth_small_device, restated with numpy"""
        import numpy as np
        from halo2_zkcert_amd.prover import splitmix64
        with np.errstate(over="ignore"):
            h = splitmix64(np.uint64(seed) + np.arange(n, dtype=np.uint64) * np.uint64(0x2545F4914F6CDD1D))
        lo, hi = h & np.uint64(0xFFFFFFFF), h >> np.uint64(32)
        v = np.where(lo % np.uint64(1000) < np.uint64(bits_per_mille), hi & np.uint64(1), hi & np.uint64((1 << word_bits) - 1))
        return zo.fr_arr_from_ints([int(x) for x in v])

    def gather(self, col, idx):
        return np.ascontiguousarray(col[idx])

    def put_rows(self, col, idx, vals):
        col[idx] = vals

    def permute(self, k, bf, cin, ctab, blind_in, blind_tab):
        return zo.permute_expression_pair(k, bf, cin, ctab, blind_in, blind_tab)

    def from_host(self, arr):
        return np.array(arr, dtype=np.uint64)

    def concat(self, cols):
        return np.concatenate(cols, axis=0)

    def coeff_to_lagrange(self, cols):
        for c in cols:
            c[:] = self.domain.coeff_to_lagrange(c, self.threads)

    def lincomb(self, polys, coeffs, low):
        return zo.linear_combination(polys, self.fr_many(coeffs), self.fr_many(low) if low else None)

    def divide_by_linear(self, srcs, roots):
        return [zo.kate_division(q, self.fr_many([r])) for q, r in zip(srcs, roots)]

    def kate_division(self, polys, roots):
        for q, rs in zip(polys, roots):
            q[:] = zo.kate_division(q, self.fr_many(rs))

    def clone(self, cols):
        return [c.copy() for c in cols]

    def partial_commit(self, cols, lagrange, first, count):
        flags = list(lagrange) if isinstance(lagrange, (list, tuple)) else [bool(lagrange)] * len(cols)
        return np.stack([zo.best_multiexp(c[first:first + count], (self.g_lagrange if f else self.g)[first:first + count], self.threads)
                         for c, f in zip(cols, flags)])

    def g1_add(self, a, b):
        o = zo.new(12)
        zo.lib().zko_g1_add(zo.p(zo.u64(a)), zo.p(zo.u64(b)), zo.p(o))
        return o

    def finish(self, jac_rows):
        res = []
        for j in range(jac_rows.shape[0]):
            a = zo.g1_to_affine(jac_rows[j])
            res.append((a, zo.g1_to_bytes(a)))
        return res

    def commit(self, cols, lagrange):
        if not cols:
            return []
        return self.finish(self.partial_commit(cols, lagrange, 0, cols[0].shape[0]))

    def commit_begin(self, cols, lagrange):
        return self.commit(cols, lagrange)

    def commit_end(self, token):
        return token

    def lagrange_to_coeff(self, cols):
        for c in cols:
            c[:] = self.domain.lagrange_to_coeff(c, self.threads)

    def coeff_to_extended(self, cols):
        return [self.domain.coeff_to_extended(c, self.threads) for c in cols]

    def evaluate_h(self, kw):
        pack = zo.EvalhPack()
        pack.build(**kw)
        return zo.evaluate_h(pack, self.domain.extended_n, self.threads)

    def evaluate_h_rows(self, kw, first_row, n_rows):
        """the row range of the full sweep (the oracle has no partial entry point: it is the checker, not the shard)"""
        return self.evaluate_h(kw)[first_row:first_row + n_rows].copy()

    def divide_and_to_coeff(self, h):
        h = self.domain.divide_by_vanishing_poly(h)
        return self.domain.extended_to_coeff_full(h, self.threads)

    def split(self, h, n, pieces):
        return [h[i * n:(i + 1) * n] for i in range(pieces)]

    def to_host(self, col):
        return col

    def compress(self, graph, fixed_l, advice_l, instance_l, theta, k):
        zero = np.zeros(4, dtype=np.uint64)
        kw = dict(k=k, extended_k=k, cs_degree=3, blinding_factors=0, extended_omega=zero, g_coset=zero, delta=zero, beta=zero,
                  gamma=zero, theta=self.fr(theta), y=zero, fixed=fixed_l, advice=advice_l, instance=instance_l, challenges=[],
                  l0=zero, l_last=zero, l_active=zero, gates_graph=graph, perm_columns=[], sigma=[], perm_z=[], lookup_graphs=[],
                  lookup_z=[], lookup_a=[], lookup_s=[], to_mont=self.fr_many)
        pack = zo.EvalhPack()
        pack.build(**kw)
        return zo.evaluate_h(pack, 1 << k, self.threads)

    def permutation_products(self, k, values, sigmas, chunk_len, beta, gamma, bf, blinding):
        nsets = -(-len(values) // chunk_len)
        return zo.permutation_products(k, values, sigmas, chunk_len, self.fr(beta), self.fr(gamma), bf, blinding.reshape(nsets, bf, 4))

    def lookup_product(self, k, cin, ctab, pin, ptab, beta, gamma, bf, blinding):
        return zo.lookup_product(k, cin, ctab, pin, ptab, self.fr(beta), self.fr(gamma), bf, blinding)

    def eval_polys_at(self, polys, xs):
        return np.stack([zo.eval_polynomial(q, self.fr(x)) for q, x in zip(polys, xs)]) if polys else np.zeros((0, 4), dtype=np.uint64)

    def grand_products(self, k, beta, gamma, bf, values, sigmas, chunk_len, perm_blinding, lookups, lookup_blinding):
        pz = self.permutation_products(k, values, sigmas, chunk_len, beta, gamma, bf, perm_blinding) if values else []
        lb = lookup_blinding.reshape(-1, bf, 4) if lookups else None
        lz = [self.lookup_product(k, *lookups[i], beta, gamma, bf, lb[i]) for i in range(len(lookups))]
        return pz, lz

    def to_host_many(self, arrays):
        return list(arrays)

    def l_cosets(self, blinding_factors):
        return self.domain.l_cosets(blinding_factors, self.threads)
