// shplonk.hip — the polynomial arithmetic of the SHPLONK multi-open prover (SURVEY.md §8 row a8 / §8(f) "next"):
// linear combinations of coefficient-form polynomials and division by prod (X - r).
//
// Restates halo2_proofs poly/kzg/multiopen/shplonk/prover.rs (create_proof: quotient_contribution,
// linearisation_contribution) and arithmetic.rs kate_division [UPSTREAM-RECALL; crate pinned at
// /root/reference/Cargo.lock:1320-1322; reached through gen_snark_shplonk, /root/reference/src/helpers.rs:233,299].
//
// Both are linear in the polynomial, so coefficients are processed in their ABI scale with the scalars in R' form.
//   * linear combination: one thread per coefficient index, all polynomials streamed once (HBM-bound: 32 B per term).
//   * kate division by (X - r): q[j] = s[j+1], s[j] = a[j] + r s[j+1] is a suffix scan of affine maps with constant
//     slope, done in three launches (tile totals, scan of the totals, rescan with carries); several polynomials, each
//     with its own root, go through one launch (grid.y).
#include <algorithm>
#include <vector>

#include "common.hpp"
using namespace zk;

#define SP_PER 8
#define SP_BLOCK 256
#define SP_TILE (SP_PER * SP_BLOCK)
#define LC_MAX 64   // polynomials per linear-combination launch

// ------------------------------------------------------------------ linear combination
// Pointers, scalars and the low-degree correction travel as kernel arguments (2.9 KB): no staging copy, no host sync.
#define LC_LOW_MAX 8
struct LcArgs {
    const uint32_t* p[LC_MAX];
    fe32 c[LC_MAX];        // R' form, canonical
    fe32 low[LC_LOW_MAX];  // ABI scale
};
// out[i] = (accumulate ? out[i] : 0) + sum_j c_j polys[j][i] - (i < nlow ? low[i] : 0)
__global__ void __launch_bounds__(256) k_lincomb(const LcArgs A, uint32_t npolys, size_t n, uint32_t nlow, int accumulate, uint32_t* out) {
    __shared__ fe sc[LC_MAX];
    if (threadIdx.x < npolys) sc[threadIdx.x] = fe_split<0>(A.c[threadIdx.x]);
    __syncthreads();
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    el1<Fr> acc = zero<Fr>();
    if (accumulate) acc = load_raw<Fr>(out + i * 8);
    uint32_t j = 0;
    for (; j + 4 <= npolys; j += 4) {
        auto t = load_raw<Fr>(A.p[j] + i * 8) * el1<Fr>(sc[j]) + load_raw<Fr>(A.p[j + 1] + i * 8) * el1<Fr>(sc[j + 1]) +
                 load_raw<Fr>(A.p[j + 2] + i * 8) * el1<Fr>(sc[j + 2]) + load_raw<Fr>(A.p[j + 3] + i * 8) * el1<Fr>(sc[j + 3]);
        acc = canonical(acc + t);
    }
    for (; j < npolys; ++j) acc = canonical(acc + load_raw<Fr>(A.p[j] + i * 8) * el1<Fr>(sc[j]));
    if (i < nlow) acc = canonical(acc - el1<Fr>(fe_split<0>(A.low[i])));
    store_raw<Fr>(out + i * 8, acc);
}

// ------------------------------------------------------------------ division by X - r
#define KD_MAX 16   // polynomials per launch
struct KdEntry {
    const uint32_t* src;
    uint32_t* dst;   // may equal src
    fe32 r;          // root, R' form canonical
};
struct KdArgs { KdEntry e[KD_MAX]; };

// suffix scan over the 256 per-thread values in LDS: S_t = sum_{q >= t} m^(q - t) A_q, m = the slope of one thread's span
__device__ __forceinline__ void suffix_scan_256(fe* sc, uint32_t t, el2<Fr> m) {
    for (uint32_t d = 1; d < SP_BLOCK; d <<= 1) {
        bool on = t + d < SP_BLOCK;
        fe mine = sc[t];
        fe other = on ? sc[t + d] : fe_zero();
        __syncthreads();
        if (on) sc[t] = canonical(el1<Fr>(mine) + el1<Fr>(other) * m).v;
        m = sqr(m);
        __syncthreads();
    }
}

// pass 1: tot[entry][blk] = the tile's Horner value at its first index with zero carry-in
__global__ void __launch_bounds__(SP_BLOCK) k_kd_totals(const KdArgs K, size_t n, uint32_t nblk, uint32_t* tot_all) {
    __shared__ fe sc[SP_BLOCK];
    const uint32_t t = threadIdx.x, e = blockIdx.y;
    const uint32_t* a = K.e[e].src;
    const el1<Fr> r(fe_split<0>(K.e[e].r));
    const size_t lo = (size_t)blockIdx.x * SP_TILE + (size_t)t * SP_PER;
    el<Fr, 4 * U> s = zero<Fr>();
#pragma unroll
    for (int j = SP_PER - 1; j >= 0; --j) {
        el1<Fr> aj = zero<Fr>();
        if (lo + j < n) aj = load_raw<Fr>(a + (lo + j) * 8);
        s = s * r + aj;
    }
    sc[t] = canonical(s).v;
    el2<Fr> m = r;
#pragma unroll
    for (int q = 0; q < 3; ++q) m = sqr(m);   // r^8
    __syncthreads();
    suffix_scan_256(sc, t, m);
    if (t == 0) mem_store(tot_all + ((size_t)e * nblk + blockIdx.x) * 8, fe_pack(sc[0]));
}
// pass 2 (one block per entry): carry[entry][blk] = s at the first index of tile blk + 1 (exclusive suffix scan, slope r^2048)
__global__ void __launch_bounds__(SP_BLOCK) k_kd_carries(const KdArgs K, uint32_t nblk, const uint32_t* tot_all, uint32_t* carry_all) {
    __shared__ fe sc[SP_BLOCK];
    const uint32_t t = threadIdx.x, e = blockIdx.x;
    const uint32_t* tot = tot_all + (size_t)e * nblk * 8;
    uint32_t* carry = carry_all + (size_t)e * nblk * 8;
    el2<Fr> M = el1<Fr>(fe_split<0>(K.e[e].r));
    for (int q = 0; q < 11; ++q) M = sqr(M);   // r^2048
    const uint32_t c = (nblk + SP_BLOCK - 1) / SP_BLOCK;
    const uint32_t lo = min(nblk, t * c), hi = min(nblk, lo + c);
    el<Fr, 4 * U> s = zero<Fr>();
    for (uint32_t q = hi; q > lo; --q) s = s * M + load_raw<Fr>(tot + (size_t)(q - 1) * 8);
    sc[t] = canonical(s).v;
    __syncthreads();
    suffix_scan_256(sc, t, pow_u64<Fr>(M, c));
    s = zero<Fr>();
    if (t + 1 < SP_BLOCK) s = el1<Fr>(sc[t + 1]);
    for (uint32_t q = hi; q > lo; --q) {
        store_raw<Fr>(carry + (size_t)(q - 1) * 8, s);
        s = s * M + load_raw<Fr>(tot + (size_t)(q - 1) * 8);
    }
}
// pass 3: q[j] = s[j + 1] from the tile's carry
__global__ void __launch_bounds__(SP_BLOCK) k_kd_apply(const KdArgs K, size_t n, uint32_t nblk, const uint32_t* carry_all) {
    __shared__ fe sc[SP_BLOCK];
    const uint32_t t = threadIdx.x, e = blockIdx.y;
    const uint32_t* a = K.e[e].src;
    uint32_t* dst = K.e[e].dst;
    const el1<Fr> r(fe_split<0>(K.e[e].r));
    const size_t lo = (size_t)blockIdx.x * SP_TILE + (size_t)t * SP_PER;
    el1<Fr> v[SP_PER];
    el<Fr, 4 * U> s = zero<Fr>();
#pragma unroll
    for (int j = SP_PER - 1; j >= 0; --j) {
        v[j] = zero<Fr>();
        if (lo + j < n) v[j] = load_raw<Fr>(a + (lo + j) * 8);
        s = s * r + v[j];
    }
    el2<Fr> m = r;
#pragma unroll
    for (int q = 0; q < 3; ++q) m = sqr(m);   // r^8
    const el1<Fr> cb = load_raw<Fr>(carry_all + ((size_t)e * nblk + blockIdx.x) * 8);
    el1<Fr> agg = canonical(s);
    if (t == SP_BLOCK - 1) agg = canonical(agg + cb * m);   // the tile's carry enters above its last thread
    sc[t] = agg.v;
    __syncthreads();
    suffix_scan_256(sc, t, m);
    s = (t + 1 < SP_BLOCK) ? el1<Fr>(sc[t + 1]) : cb;
#pragma unroll
    for (int j = SP_PER - 1; j >= 0; --j) {
        if (lo + j < n) store_raw<Fr>(dst + (lo + j) * 8, s);
        s = s * r + v[j];
    }
}

static fe32 abi_to_raw(const uint64_t* p) { return fe_pack(fe_canonical<Fr>(from_abi<Fr>(mem_load(p)).v)); }

// dst_j = src_j / (X - root_j) for `count` polynomials, KD_MAX per launch triple
static int divide_round(zkhip_ctx* ctx, size_t n, const std::vector<KdEntry>& ents) {
    hipStream_t st = ctx->stream;
    uint32_t nblk = div_up(n, SP_TILE);
    void *d_tot, *d_carry;
    ZK_TRY(ctx->get_scratch("kd_tot", (size_t)KD_MAX * nblk * 32, &d_tot));
    ZK_TRY(ctx->get_scratch("kd_carry", (size_t)KD_MAX * nblk * 32, &d_carry));
    for (size_t done = 0; done < ents.size(); done += KD_MAX) {
        uint32_t cnt = (uint32_t)std::min<size_t>(KD_MAX, ents.size() - done);
        KdArgs K;
        memset(&K, 0, sizeof K);
        for (uint32_t j = 0; j < cnt; ++j) K.e[j] = ents[done + j];
        hipLaunchKernelGGL(k_kd_totals, dim3(nblk, cnt), dim3(SP_BLOCK), 0, st, K, n, nblk, (uint32_t*)d_tot);
        hipLaunchKernelGGL(k_kd_carries, dim3(cnt), dim3(SP_BLOCK), 0, st, K, nblk, (const uint32_t*)d_tot, (uint32_t*)d_carry);
        hipLaunchKernelGGL(k_kd_apply, dim3(nblk, cnt), dim3(SP_BLOCK), 0, st, K, n, nblk, (const uint32_t*)d_carry);
    }
    ZK_LAUNCH_CHECK();
    return ZKHIP_OK;
}

extern "C" {

int zkhip_linear_combination_device(zkhip_ctx* ctx, size_t n, const void* const* d_polys, size_t npolys, const uint64_t* coeffs,
                                    const uint64_t* low, size_t nlow, void* d_out) {
    if (!ctx || !d_out || (npolys && (!d_polys || !coeffs)) || (nlow && !low)) { set_error("zkhip_linear_combination_device: null argument"); return ZKHIP_EINVAL; }
    if (nlow > n || nlow > LC_LOW_MAX) { set_error("zkhip_linear_combination_device: nlow = %zu unsupported (<= min(n, %d))", nlow, LC_LOW_MAX); return ZKHIP_EINVAL; }
    if (n == 0) return ZKHIP_OK;
    ProfScope ps(ctx, "linear_combination");
    size_t done = 0;
    do {
        uint32_t cnt = (uint32_t)std::min<size_t>(LC_MAX, npolys - done);
        bool last = done + cnt == npolys;
        LcArgs A;
        memset(&A, 0, sizeof A);
        for (uint32_t j = 0; j < cnt; ++j) {
            A.p[j] = (const uint32_t*)d_polys[done + j];
            A.c[j] = abi_to_raw(coeffs + 4 * (done + j));
        }
        if (last) for (size_t j = 0; j < nlow; ++j) A.low[j] = mem_load(low + 4 * j);
        hipLaunchKernelGGL(k_lincomb, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, A, cnt, n, last ? (uint32_t)nlow : 0u, done ? 1 : 0,
                           (uint32_t*)d_out);
        done += cnt;
    } while (done < npolys);
    ZK_LAUNCH_CHECK();
    return ZKHIP_OK;
}

int zkhip_divide_by_linear_device(zkhip_ctx* ctx, size_t n, const void* const* d_src, void* const* d_dst, size_t npolys, const uint64_t* roots) {
    if (!ctx || (npolys && (!d_src || !d_dst || !roots))) { set_error("zkhip_divide_by_linear_device: null argument"); return ZKHIP_EINVAL; }
    if (npolys == 0 || n == 0) return ZKHIP_OK;
    std::vector<KdEntry> ents(npolys);
    for (size_t j = 0; j < npolys; ++j) {
        ents[j].src = (const uint32_t*)d_src[j];
        ents[j].dst = (uint32_t*)d_dst[j];
        ents[j].r = abi_to_raw(roots + 4 * j);
    }
    ProfScope ps(ctx, "kate_division");
    return divide_round(ctx, n, ents);
}

int zkhip_kate_division_device(zkhip_ctx* ctx, size_t n, void* const* d_polys, size_t npolys, const uint32_t* nroots, const uint64_t* roots) {
    if (!ctx || (npolys && (!d_polys || !nroots))) { set_error("zkhip_kate_division_device: null argument"); return ZKHIP_EINVAL; }
    if (npolys == 0 || n == 0) return ZKHIP_OK;
    uint32_t max_roots = 0;
    size_t total = 0;
    for (size_t j = 0; j < npolys; ++j) { max_roots = std::max(max_roots, nroots[j]); total += nroots[j]; }
    if (total && !roots) { set_error("zkhip_kate_division_device: null roots"); return ZKHIP_EINVAL; }
    ProfScope ps(ctx, "kate_division");
    // round t divides every polynomial that has more than t roots by its t-th root
    for (uint32_t t = 0; t < max_roots; ++t) {
        std::vector<KdEntry> ents;
        size_t off = 0;
        for (size_t j = 0; j < npolys; ++j) {
            if (nroots[j] > t) { KdEntry e; e.src = (const uint32_t*)d_polys[j]; e.dst = (uint32_t*)d_polys[j]; e.r = abi_to_raw(roots + 4 * (off + t)); ents.push_back(e); }
            off += nroots[j];
        }
        ZK_TRY(divide_round(ctx, n, ents));
    }
    return ZKHIP_OK;
}

}  // extern "C"
