// polyops.hip — the O(n) field work of create_proof between the commitments ("next" tier):
// batch inversion, running products (permutation / lookup grand products), polynomial evaluation at a point.
//
// Restates halo2_proofs plonk/permutation/prover.rs (Argument::commit), plonk/lookup/prover.rs (commit_product),
// arithmetic.rs (eval_polynomial) and ff::BatchInvert [UPSTREAM-RECALL; crate pinned at
// /root/reference/Cargo.lock:1320-1322].  Values are unique, so the evaluation order is free; the random blinding
// rows upstream draws from its rng are an input.
//
// All three are latency-shaped on a GPU (a Fermat inversion is 380 dependent products; a scan is log-depth), so the
// kernels keep 8 elements per thread in registers, combine 256 threads through an LDS product tree and use ONE
// inversion per 2048 elements.
#include <algorithm>

#include "common.hpp"
using namespace zk;

#define PO_PER 8
#define PO_BLOCK 256
#define PO_TILE (PO_PER * PO_BLOCK)

// ------------------------------------------------------------------ batch inversion
// FORM 0: raw R'-form scratch (internal);  FORM 1: ABI values (x 2^256) in and out.
template <int FORM>
__device__ __forceinline__ el2<Fr> inv_load(const uint32_t* p) {
    if (FORM == 0) return load_raw<Fr>(p);
    return reduce(load_x32<Fr>(p));
}
template <int FORM>
__device__ __forceinline__ void inv_store(uint32_t* p, const el2<Fr>& v) {
    if (FORM == 0) store_raw<Fr>(p, v); else store_div32<Fr>(p, v);
}

// LEVEL 0: the tile's root is inverted here (one binary-Euclid inversion per 2048 elements: ~30 k instructions on one wave, as much VALU
//          work as everything else in the tile).
// LEVEL 1: the tile's root product is only STORED (roots[tile], raw R' form); nothing is written to `a`.
// LEVEL 2: the root's inverse is READ from roots[tile] (a second-level batch inversion of all roots ran in between).
// Two-level use (launch_batch_invert, n >= 2^16): level 1, k_batch_invert<0, 0> over the roots, level 2 — one inversion per 2^22
// elements instead of 2048 of them, for one extra pass over the data.
template <int FORM, int LEVEL = 0>
__global__ void __launch_bounds__(PO_BLOCK) k_batch_invert(uint32_t* a, size_t n, uint32_t* roots = nullptr) {
    __shared__ fe tree[2 * PO_BLOCK];   // tree[256 + t] = leaf t, tree[i] = tree[2i] * tree[2i+1]
    const uint32_t t = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * PO_TILE;
    // element j of thread t: index base + j * 256 + t (coalesced; the grouping is irrelevant to an inversion)
    el2<Fr> v[PO_PER], pre[PO_PER];
    el2<Fr> acc = one<Fr>();
#pragma unroll
    for (int j = 0; j < PO_PER; ++j) {
        size_t i = base + (size_t)j * PO_BLOCK + t;
        v[j] = one<Fr>();
        if (i < n) {
            v[j] = inv_load<FORM>(a + i * 8);
            if (fe_is_zero_modp<Fr>(v[j].v)) v[j] = el2<Fr>(fe_zero());
        }
        pre[j] = acc;
        if (!fe_is_zero_exact(v[j].v)) acc = acc * v[j];
    }
    tree[PO_BLOCK + t] = acc.v;
    __syncthreads();
    for (uint32_t w = PO_BLOCK / 2; w >= 1; w >>= 1) {
        if (t < w) tree[w + t] = (el2<Fr>(tree[2 * (w + t)]) * el2<Fr>(tree[2 * (w + t) + 1])).v;
        __syncthreads();
    }
    if (LEVEL == 1) {
        // an all-zero tile has root 1 (zeros are skipped in the products), never 0: the second level needs no special case
        if (t == 0) store_raw<Fr>(roots + (size_t)blockIdx.x * 8, el2<Fr>(tree[1]));
        return;
    }
    // every lane of wave 0 inverts the root (same cost as one lane; avoids a broadcast)
    __shared__ fe root_inv;
    if (LEVEL == 2) {
        if (t == 0) root_inv = load_raw<Fr>(roots + (size_t)blockIdx.x * 8).v;
    } else if (t < 64) {
        el2<Fr> r = inv_euclid<Fr>(el2<Fr>(tree[1]));   // one value, the same in all 64 lanes: binary Euclid beats the 380-product Fermat chain
        if (t == 0) root_inv = r.v;
    }
    __syncthreads();
    // down-sweep: inv(node) known -> inv(left) = inv(node) * right, inv(right) = inv(node) * left
    if (t == 0) tree[1] = root_inv;
    __syncthreads();
    for (uint32_t w = 1; w < PO_BLOCK; w <<= 1) {
        if (t < w) {
            uint32_t node = w + t;
            el2<Fr> in(tree[node]), l(tree[2 * node]), r(tree[2 * node + 1]);
            tree[2 * node] = (in * r).v;
            tree[2 * node + 1] = (in * l).v;
        }
        __syncthreads();
    }
    el2<Fr> iv(tree[PO_BLOCK + t]);   // inverse of this thread's product
#pragma unroll
    for (int j = PO_PER - 1; j >= 0; --j) {
        size_t i = base + (size_t)j * PO_BLOCK + t;
        if (!fe_is_zero_exact(v[j].v)) {
            el2<Fr> r = iv * pre[j];
            iv = iv * v[j];
            if (i < n) inv_store<FORM>(a + i * 8, r);
        }
    }
}

// in place over n elements of form FORM on the context's stream
template <int FORM>
static int launch_batch_invert(zkhip_ctx* ctx, void* d_a, size_t n) {
    const unsigned tiles = div_up(n, PO_TILE);
    if (n < ((size_t)1 << 16)) {
        hipLaunchKernelGGL((k_batch_invert<FORM, 0>), dim3(tiles), dim3(PO_BLOCK), 0, ctx->stream, (uint32_t*)d_a, n, (uint32_t*)nullptr);
    } else {
        void* d_roots;
        ZK_TRY(ctx->get_scratch("po_inv_roots", (size_t)tiles * 32, &d_roots));
        hipLaunchKernelGGL((k_batch_invert<FORM, 1>), dim3(tiles), dim3(PO_BLOCK), 0, ctx->stream, (uint32_t*)d_a, n, (uint32_t*)d_roots);
        hipLaunchKernelGGL((k_batch_invert<0, 0>), dim3(div_up(tiles, PO_TILE)), dim3(PO_BLOCK), 0, ctx->stream, (uint32_t*)d_roots, (size_t)tiles, (uint32_t*)nullptr);
        hipLaunchKernelGGL((k_batch_invert<FORM, 2>), dim3(tiles), dim3(PO_BLOCK), 0, ctx->stream, (uint32_t*)d_a, n, (uint32_t*)d_roots);
    }
    ZK_LAUNCH_CHECK();
    return ZKHIP_OK;
}

// ------------------------------------------------------------------ running products
// Exclusive prefix products over `nseg` independent arrays of n raw R'-form values (in), 8 consecutive per thread:
//   pass 1: out_local[i] = product of in[tile_start .. i) within the tile, block_tot[seg][blk] = product of the tile
//   pass 2: block_pre[seg][blk] = product of the tiles before blk (one block per segment)
//   pass 3: z[i] = first[seg] * block_pre * out_local   (ABI form), rows >= n_keep replaced by the blinding values
__global__ void __launch_bounds__(PO_BLOCK) k_rp_local(const uint32_t* in_all, size_t n, uint32_t* local_all, uint32_t* block_tot_all,
                                                       uint32_t nblk) {
    __shared__ fe sc[PO_BLOCK];
    const uint32_t t = threadIdx.x, seg = blockIdx.y;
    const uint32_t* in = in_all + (size_t)seg * n * 8;
    uint32_t* local = local_all + (size_t)seg * n * 8;
    const size_t lo = (size_t)blockIdx.x * PO_TILE + (size_t)t * PO_PER;
    el2<Fr> pre[PO_PER];
    el2<Fr> acc = one<Fr>();
#pragma unroll
    for (int j = 0; j < PO_PER; ++j) {
        pre[j] = acc;
        if (lo + j < n) acc = acc * load_raw<Fr>(in + (lo + j) * 8);
    }
    // inclusive scan of the per-thread products (Hillis-Steele, 8 steps)
    sc[t] = acc.v;
    __syncthreads();
    for (uint32_t d = 1; d < PO_BLOCK; d <<= 1) {
        fe other;
        bool has = t >= d;
        if (has) other = sc[t - d];
        __syncthreads();
        if (has) sc[t] = (el2<Fr>(sc[t]) * el2<Fr>(other)).v;
        __syncthreads();
    }
    el2<Fr> before = t ? el2<Fr>(sc[t - 1]) : el2<Fr>(one<Fr>());
#pragma unroll
    for (int j = 0; j < PO_PER; ++j)
        if (lo + j < n) store_raw<Fr>(local + (lo + j) * 8, before * pre[j]);
    if (t == PO_BLOCK - 1) store_raw<Fr>(block_tot_all + ((size_t)seg * nblk + blockIdx.x) * 8, el2<Fr>(sc[t]));
}
// seg_total (optional): the product of ALL of the segment's tiles — a row-range shard's contribution to the ranks above it
__global__ void __launch_bounds__(1024) k_rp_blocks(const uint32_t* block_tot_all, uint32_t* block_pre_all, uint32_t nblk, uint32_t* seg_total = nullptr) {
    __shared__ fe sc[1024];
    const uint32_t t = threadIdx.x, seg = blockIdx.x;
    const uint32_t* tot = block_tot_all + (size_t)seg * nblk * 8;
    uint32_t* pre = block_pre_all + (size_t)seg * nblk * 8;
    el2<Fr> carry = one<Fr>();
    for (uint32_t b0 = 0; b0 < nblk; b0 += 1024) {
        uint32_t b = b0 + t;
        sc[t] = b < nblk ? load_raw<Fr>(tot + (size_t)b * 8).v : one<Fr>().v;
        __syncthreads();
        for (uint32_t d = 1; d < 1024; d <<= 1) {
            fe other;
            bool has = t >= d;
            if (has) other = sc[t - d];
            __syncthreads();
            if (has) sc[t] = (el2<Fr>(sc[t]) * el2<Fr>(other)).v;
            __syncthreads();
        }
        el2<Fr> excl = t ? el2<Fr>(sc[t - 1]) : el2<Fr>(one<Fr>());
        if (b < nblk) store_raw<Fr>(pre + (size_t)b * 8, carry * excl);
        carry = carry * el2<Fr>(sc[1023]);
        __syncthreads();
    }
    if (seg_total && t == 0) store_raw<Fr>(seg_total + (size_t)seg * 8, carry);
}
// Row-range shards (one proof over several GPUs, prover.hip): rank R holds rows [R m, (R + 1) m) of every segment.  Each rank publishes,
// per segment, the product of its rows' terms (k_rp_blocks' seg_total) and — the rank that holds the chain row — its local unchained
// prefix there (k_rp_pick); after the all-gather k_rp_shard_first turns them into this rank's multiplier of the whole segment:
// (product of the lower ranks' totals) x (the chain: product over the earlier chained segments of their FULL prefix at the chain row).
__global__ void k_rp_pick(const uint32_t* local_all, const uint32_t* block_pre_all, size_t n, uint32_t nblk, uint32_t nseg, size_t row, int have,
                          uint32_t* out) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nseg) return;
    el2<Fr> v = one<Fr>();
    if (have) v = load_raw<Fr>(local_all + ((size_t)s * n + row) * 8) * load_raw<Fr>(block_pre_all + ((size_t)s * nblk + row / PO_TILE) * 8);
    store_raw<Fr>(out + (size_t)s * 8, v);
}
// xchg[rank][0..nseg) = totals, xchg[rank][nseg..2 nseg) = picks
__global__ void k_rp_shard_first(const uint32_t* xchg, uint32_t nranks, uint32_t rank, uint32_t nseg, uint32_t nchain, uint32_t chain_rank,
                                 uint32_t* seg_first) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    el2<Fr> f = one<Fr>();
    for (uint32_t s = 0; s < nseg; ++s) {
        el2<Fr> below = one<Fr>();
        for (uint32_t r = 0; r < rank; ++r) below = below * load_raw<Fr>(xchg + ((size_t)r * 2 * nseg + s) * 8);
        const bool chained = s < nchain && nchain > 1;
        el2<Fr> first = below;
        if (chained) first = below * f;
        store_raw<Fr>(seg_first + (size_t)s * 8, first);
        if (chained) {   // the segment's full unchained prefix at the chain row
            el2<Fr> last = load_raw<Fr>(xchg + ((size_t)chain_rank * 2 * nseg + nseg + s) * 8);
            for (uint32_t r = 0; r < chain_rank; ++r) last = last * load_raw<Fr>(xchg + ((size_t)r * 2 * nseg + s) * 8);
            f = f * last;
        }
    }
}
// seg_first[seg]: raw R'-form multiplier of the whole segment (chains the permutation sets); may be null (= 1)
__global__ void __launch_bounds__(PO_BLOCK) k_rp_apply(const uint32_t* local_all, const uint32_t* block_pre_all, const uint32_t* seg_first,
                                                       size_t n, uint32_t nblk, uint32_t* const* z_out, size_t n_keep,
                                                       const uint32_t* blinding_all /* [seg][n - n_keep] ABI */) {
    const uint32_t seg = blockIdx.y;
    size_t i = (size_t)blockIdx.x * PO_BLOCK + threadIdx.x;
    if (i >= n) return;
    uint32_t* z = z_out[seg];
    if (i >= n_keep) {
        mem_store(z + i * 8, mem_load(blinding_all + ((size_t)seg * (n - n_keep) + (i - n_keep)) * 8));
        return;
    }
    el2<Fr> v = load_raw<Fr>(local_all + ((size_t)seg * n + i) * 8) * load_raw<Fr>(block_pre_all + ((size_t)seg * nblk + i / PO_TILE) * 8);
    if (seg_first) v = v * load_raw<Fr>(seg_first + (size_t)seg * 8);
    store_div32<Fr>(z + i * 8, v);   // R' form -> ABI form
}
// seg_first[0] = 1, seg_first[s] = seg_first[s-1] * (unchained z_{s-1}[last_row])  with unchained z = block_pre * local
// (segments >= nchain are independent: first = 1)
__global__ void k_rp_chain(const uint32_t* local_all, const uint32_t* block_pre_all, size_t n, uint32_t nblk, uint32_t nseg, uint32_t nchain,
                           size_t last_row, uint32_t* seg_first) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    el2<Fr> f = one<Fr>();
    for (uint32_t s = 0; s < nseg; ++s) {
        if (s >= nchain) { store_raw<Fr>(seg_first + (size_t)s * 8, one<Fr>()); continue; }
        store_raw<Fr>(seg_first + (size_t)s * 8, f);
        el2<Fr> last = load_raw<Fr>(local_all + ((size_t)s * n + last_row) * 8) * load_raw<Fr>(block_pre_all + ((size_t)s * nblk + last_row / PO_TILE) * 8);
        f = f * last;
    }
}

// ------------------------------------------------------------------ permutation / lookup fractions
struct PermParams {
    const uint32_t* const* values;   // ncols ABI columns (Lagrange)
    const uint32_t* const* sigmas;   // ncols ABI columns
    uint32_t ncols, chunk_len;
    fe beta, gamma;                  // R' form, canonical
    const uint32_t* dbeta;           // [ncols] raw R' form: delta^j * beta
    const uint32_t* w_lo; const uint32_t* w_hi; uint32_t w_h;   // omega^i = lo * hi
};
// DEN = 1: out[set][i] = prod_j (beta sigma_j + gamma + v_j);  DEN = 0: out[set][i] *= prod_j (delta^j w^i beta + gamma + v_j)
template <int DEN>
__global__ void __launch_bounds__(256) k_perm_terms(PermParams P, size_t n, uint32_t* out_all) {
    const uint32_t set = blockIdx.y;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const el1<Fr> beta(P.beta), gamma(P.gamma);
    uint32_t c0 = set * P.chunk_len, c1 = min(c0 + P.chunk_len, P.ncols);
    uint32_t* out = out_all + ((size_t)set * n + i) * 8;
    el2<Fr> acc = one<Fr>();
    if (DEN) {
        for (uint32_t c = c0; c < c1; ++c)
            acc = acc * (beta * load_x32<Fr>(P.sigmas[c] + i * 8) + gamma + load_x32<Fr>(P.values[c] + i * 8));
    } else {
        acc = load_raw<Fr>(out);
        el2<Fr> wi = load_raw<Fr>(P.w_lo + (i & (((size_t)1 << P.w_h) - 1)) * 8) * load_raw<Fr>(P.w_hi + (i >> P.w_h) * 8);
        for (uint32_t c = c0; c < c1; ++c)
            acc = acc * (wi * load_raw<Fr>(P.dbeta + (size_t)c * 8) + gamma + load_x32<Fr>(P.values[c] + i * 8));
    }
    store_raw<Fr>(out, acc);
}
// DEN = 1: out[i] = (pin + beta)(ptab + gamma);  DEN = 0: out[i] *= (cin + beta)(ctab + gamma)
template <int DEN>
__global__ void __launch_bounds__(256) k_lookup_terms(const uint32_t* a, const uint32_t* b, fe beta_v, fe gamma_v, size_t n, uint32_t* out) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const el1<Fr> beta(beta_v), gamma(gamma_v);
    auto t = (load_x32<Fr>(a + i * 8) + beta) * (load_x32<Fr>(b + i * 8) + gamma);
    if (DEN) store_raw<Fr>(out + i * 8, t);
    else store_raw<Fr>(out + i * 8, load_raw<Fr>(out + i * 8) * t);
}

// ------------------------------------------------------------------ evaluation at a point
// partial[poly][blk] = sum_{i in tile} c_i x^(i - tile_start)   (coefficients loaded raw: the sum is linear in them)
// xs: raw R'-form points, one per polynomial
// (polynomial, point) pairs of a launch: by value in the kernel arguments for <= EV_MAX pairs (dev pointers null), so that the
// evaluations — on the critical path right after the challenge x — need no upload
#define EV_MAX 48
struct EvArgs { const uint32_t* const* polys; const uint32_t* xs; const uint32_t* p[EV_MAX]; fe32 x[EV_MAX]; };
__device__ __forceinline__ const uint32_t* ev_poly(const EvArgs& A, uint32_t i) { return A.polys ? A.polys[i] : A.p[i]; }
__device__ __forceinline__ el1<Fr> ev_x(const EvArgs& A, uint32_t i) { return A.xs ? load_raw<Fr>(A.xs + (size_t)i * 8) : el1<Fr>(fe_split<0>(A.x[i])); }
// PER coefficients per thread: the block tree (8 levels of a product and a squaring by every thread) is a fixed cost per thread, so
// long polynomials use 32 (1.5 products per coefficient instead of 2.6)
template <int PER>
__global__ void __launch_bounds__(PO_BLOCK) k_eval_tiles(const EvArgs A, size_t n, uint32_t* partial_all, uint32_t nblk) {
    __shared__ fe sc[PO_BLOCK];
    __shared__ fe xpw[9];   // x^(2^l), l = 0 .. 8; xpw[8] = x^256
    const uint32_t t = threadIdx.x, poly = blockIdx.y;
    const uint32_t* c = ev_poly(A, poly);
    const el2<Fr> x = ev_x(A, poly);
    // The powers every thread needs are the same: thread 0 squares them once into LDS (instead of 256 threads squaring along).
    if (t == 0) {
        el2<Fr> xp = x;
#pragma unroll
        for (int l = 0; l < 9; ++l) { xpw[l] = xp.v; xp = sqr(xp); }
    }
    __syncthreads();
    // Thread t owns coefficients t, t + 256, t + 512, ... of the tile (Horner in y = x^256): the 64 lanes of a wave read 2 KiB of
    // consecutive bytes per step.  (Until round 4 a thread owned PER CONSECUTIVE coefficients — 1 KiB per lane, every load of a wave
    // touching 64 different lines: 1.7 TB/s at 2^22; the tile's value is the same sum either way.)
    const el2<Fr> y(xpw[8]);
    const size_t base = (size_t)blockIdx.x * (PER * PO_BLOCK) + t;
    el<Fr, 4 * U> acc = zero<Fr>();
#pragma unroll
    for (int j = PER - 1; j >= 0; --j) {
        el1<Fr> cj = zero<Fr>();
        const size_t i = base + (size_t)j * PO_BLOCK;
        if (i < n) cj = load_raw<Fr>(c + i * 8);
        acc = acc * y + cj;
    }
    sc[t] = acc.v;   // P_t(y) = sum_j c[base + 256 j] y^j; the tile's value is sum_t x^t P_t(y)
    __syncthreads();
    // pairwise: left + right * x^span
    int lvl = 0;
    for (uint32_t span = 1; span < PO_BLOCK; span <<= 1, ++lvl) {
        if ((t & (2 * span - 1)) == 0) sc[t] = (el<Fr, 4 * U>(sc[t]) + el<Fr, 4 * U>(sc[t + span]) * el2<Fr>(xpw[lvl])).v;   // < 4p + 2p: contract below
        __syncthreads();
        if ((t & (2 * span - 1)) == 0) sc[t] = reduce(el<Fr, 8 * U>(sc[t])).v;
        __syncthreads();
    }
    if (t == 0) store_raw<Fr>(partial_all + ((size_t)poly * nblk + blockIdx.x) * 8, el2<Fr>(sc[0]));
}
// out[poly] = sum_b partial[poly][b] * x^(tile b), one block per polynomial
__global__ void __launch_bounds__(PO_BLOCK) k_eval_final(const EvArgs A, const uint32_t* partial_all, uint32_t nblk, uint32_t* out, uint32_t tile) {
    __shared__ fe sc[PO_BLOCK];
    const uint32_t t = threadIdx.x, poly = blockIdx.x;
    const uint32_t* part = partial_all + (size_t)poly * nblk * 8;
    el2<Fr> xt = pow_u64<Fr>(el2<Fr>(ev_x(A, poly)), tile);   // x^tile
    el2<Fr> step = pow_u64<Fr>(xt, PO_BLOCK);                        // x^(2048 * 256)
    el2<Fr> xw = pow_u64<Fr>(xt, t);                                 // x^(2048 t)
    el<Fr, 4 * U> acc = zero<Fr>();
    for (uint32_t b = t; b < nblk; b += PO_BLOCK) {
        acc = reduce(acc + load_raw<Fr>(part + (size_t)b * 8) * xw);
        xw = xw * step;
    }
    sc[t] = acc.v;
    __syncthreads();
    for (uint32_t d = PO_BLOCK / 2; d >= 1; d >>= 1) {
        if (t < d) sc[t] = reduce(el<Fr, 4 * U>(sc[t]) + el<Fr, 4 * U>(sc[t + d])).v;
        __syncthreads();
    }
    if (t == 0) store_raw<Fr>(out + (size_t)poly * 8, el2<Fr>(sc[0]));
}

// ------------------------------------------------------------------ host drivers
static fe32 abi_to_raw(const uint64_t* p) { return fe_pack(fe_canonical<Fr>(from_abi<Fr>(mem_load(p)).v)); }
static fe abi_to_fe(const uint64_t* p) { return fe_canonical<Fr>(from_abi<Fr>(mem_load(p)).v); }

// z_out[seg] (ABI) = running products of nseg raw arrays `d_terms` ([seg][n]); chain: multiply segment s by the last kept row of s-1.
// shard != nullptr: the arrays are rows [row0, row0 + n) of n_total (this rank's range; n_keep and chain_row are GLOBAL rows then).
struct RpShard { size_t row0, n_total; };
static int running_products(zkhip_ctx* ctx, const void* d_terms, uint32_t nseg, size_t n, size_t n_keep, uint32_t nchain, size_t chain_row,
                            const void* d_blinding, void* const* z_out_host, const RpShard* shard = nullptr) {
    const bool chain = nchain > 1;
    hipStream_t st = ctx->stream;
    uint32_t nblk = div_up(n, PO_TILE);
    void *d_local, *d_tot, *d_pre, *d_first, *d_zptr;
    ZK_TRY(ctx->get_scratch("po_local", (size_t)nseg * n * 32, &d_local));
    ZK_TRY(ctx->get_scratch("po_tot", (size_t)nseg * nblk * 32, &d_tot));
    ZK_TRY(ctx->get_scratch("po_pre", (size_t)nseg * nblk * 32, &d_pre));
    ZK_TRY(ctx->get_scratch("po_first", (size_t)nseg * 32, &d_first));
    ZK_TRY(ctx->get_scratch("po_zptr", (size_t)nseg * sizeof(void*), &d_zptr));
    ZK_TRY(ctx->upload(d_zptr, z_out_host, nseg * sizeof(void*)));
    hipLaunchKernelGGL(k_rp_local, dim3(nblk, nseg), dim3(PO_BLOCK), 0, st, (const uint32_t*)d_terms, n, (uint32_t*)d_local, (uint32_t*)d_tot, nblk);
    if (shard) {
        const uint32_t NR = (uint32_t)ctx->comm.nranks, RK = (uint32_t)ctx->comm.rank;
        void* d_x;
        ZK_TRY(ctx->get_scratch("po_xchg", (size_t)NR * 2 * nseg * 32, &d_x));
        uint32_t* mine = (uint32_t*)d_x + (size_t)RK * 2 * nseg * 8;
        hipLaunchKernelGGL(k_rp_blocks, dim3(nseg), dim3(1024), 0, st, (const uint32_t*)d_tot, (uint32_t*)d_pre, nblk, mine);
        const uint32_t chain_rank = (uint32_t)(chain_row / n);
        const bool have = chain_row >= shard->row0 && chain_row < shard->row0 + n;
        hipLaunchKernelGGL(k_rp_pick, dim3(div_up(nseg, 64)), dim3(64), 0, st, (const uint32_t*)d_local, (const uint32_t*)d_pre, n, nblk, nseg,
                           have ? chain_row - shard->row0 : (size_t)0, have ? 1 : 0, mine + (size_t)nseg * 8);
        ZK_LAUNCH_CHECK();
        ZK_TRY(zk::comm_allgather(ctx, mine, d_x, (size_t)2 * nseg * 32));
        hipLaunchKernelGGL(k_rp_shard_first, dim3(1), dim3(64), 0, st, (const uint32_t*)d_x, NR, RK, nseg, nchain, chain_rank, (uint32_t*)d_first);
        const size_t keep_local = n_keep <= shard->row0 ? 0 : std::min(n, n_keep - shard->row0);
        hipLaunchKernelGGL(k_rp_apply, dim3(div_up(n, PO_BLOCK), nseg), dim3(PO_BLOCK), 0, st, (const uint32_t*)d_local, (const uint32_t*)d_pre,
                           (const uint32_t*)d_first, n, nblk, (uint32_t* const*)d_zptr, keep_local, (const uint32_t*)d_blinding);
        ZK_LAUNCH_CHECK();
        return ZKHIP_OK;
    }
    hipLaunchKernelGGL(k_rp_blocks, dim3(nseg), dim3(1024), 0, st, (const uint32_t*)d_tot, (uint32_t*)d_pre, nblk, (uint32_t*)nullptr);
    if (chain) hipLaunchKernelGGL(k_rp_chain, dim3(1), dim3(64), 0, st, (const uint32_t*)d_local, (const uint32_t*)d_pre, n, nblk, nseg, nchain,
                                  chain_row, (uint32_t*)d_first);
    hipLaunchKernelGGL(k_rp_apply, dim3(div_up(n, PO_BLOCK), nseg), dim3(PO_BLOCK), 0, st, (const uint32_t*)d_local, (const uint32_t*)d_pre,
                       chain ? (const uint32_t*)d_first : (const uint32_t*)nullptr, n, nblk, (uint32_t* const*)d_zptr, n_keep,
                       (const uint32_t*)d_blinding);
    ZK_LAUNCH_CHECK();
    return ZKHIP_OK;
}

extern "C" {

int zkhip_batch_invert_device(zkhip_ctx* ctx, void* d_a, size_t n) {
    if (!ctx || !d_a) { set_error("zkhip_batch_invert_device: null argument"); return ZKHIP_EINVAL; }
    if (n == 0) return ZKHIP_OK;
    ProfScope ps(ctx, "batch_invert");
    return launch_batch_invert<1>(ctx, d_a, n);
}

// xs_host: npolys ABI points (one per polynomial)
static int eval_at(zkhip_ctx* ctx, const void* const* d_polys, size_t npolys, size_t n, const uint64_t* xs_host, void* d_out) {
    if (npolys == 0) return ZKHIP_OK;
    if (n == 0) { ZK_HIP(hipMemsetAsync(d_out, 0, npolys * 32, ctx->stream)); return ZKHIP_OK; }
    const uint32_t per = n >= ((size_t)1 << 16) ? 32 : PO_PER, tile = per * PO_BLOCK;
    uint32_t nblk = div_up(n, tile);
    void *d_ptrs, *d_part, *d_xs;
    ZK_TRY(ctx->get_scratch("po_eval_ptrs", npolys * sizeof(void*), &d_ptrs));
    ZK_TRY(ctx->get_scratch("po_eval_part", npolys * (size_t)nblk * 32, &d_part));
    ZK_TRY(ctx->get_scratch("po_eval_xs", npolys * 32, &d_xs));
    std::vector<fe32> xs(npolys);
    for (size_t j = 0; j < npolys; ++j) xs[j] = abi_to_raw(xs_host + 4 * j);
    EvArgs A;
    const bool by_value = ctx->opt.eval_byval != 0;
    if (npolys <= EV_MAX && by_value) {
        A.polys = nullptr; A.xs = nullptr;
        for (size_t j = 0; j < npolys; ++j) { A.p[j] = (const uint32_t*)d_polys[j]; A.x[j] = xs[j]; }
    } else {
        ZK_TRY(ctx->upload(d_ptrs, d_polys, npolys * sizeof(void*)));
        ZK_TRY(ctx->upload(d_xs, xs.data(), npolys * 32));
        A.polys = (const uint32_t* const*)d_ptrs; A.xs = (const uint32_t*)d_xs;
    }
    ProfScope ps(ctx, "eval_polynomial");
    if (per == 32)
        hipLaunchKernelGGL(k_eval_tiles<32>, dim3(nblk, (unsigned)npolys), dim3(PO_BLOCK), 0, ctx->stream, A, n, (uint32_t*)d_part, nblk);
    else
        hipLaunchKernelGGL(k_eval_tiles<PO_PER>, dim3(nblk, (unsigned)npolys), dim3(PO_BLOCK), 0, ctx->stream, A, n, (uint32_t*)d_part, nblk);
    hipLaunchKernelGGL(k_eval_final, dim3((unsigned)npolys), dim3(PO_BLOCK), 0, ctx->stream, A, (const uint32_t*)d_part, nblk, (uint32_t*)d_out, tile);
    ZK_LAUNCH_CHECK();
    return ZKHIP_OK;
}
int zkhip_eval_polynomial_device(zkhip_ctx* ctx, const void* const* d_polys, size_t npolys, size_t n, const uint64_t x[4], void* d_out) {
    if (!ctx || !d_polys || !x || !d_out) { set_error("zkhip_eval_polynomial_device: null argument"); return ZKHIP_EINVAL; }
    std::vector<uint64_t> xs(4 * (npolys ? npolys : 1));
    for (size_t j = 0; j < npolys; ++j) memcpy(&xs[4 * j], x, 32);
    return eval_at(ctx, d_polys, npolys, n, xs.data(), d_out);
}
int zkhip_eval_polynomials_at_device(zkhip_ctx* ctx, const void* const* d_polys, size_t npolys, size_t n, const uint64_t* xs, void* d_out) {
    if (!ctx || !d_polys || !xs || !d_out) { set_error("zkhip_eval_polynomials_at_device: null argument"); return ZKHIP_EINVAL; }
    return eval_at(ctx, d_polys, npolys, n, xs, d_out);
}

int zkhip_permutation_products_device(zkhip_ctx* ctx, uint32_t k, const void* const* d_values, const void* const* d_sigmas, size_t ncols,
                                      uint32_t chunk_len, const uint64_t beta[4], const uint64_t gamma[4], uint32_t blinding_factors,
                                      const void* d_blinding, void* const* d_z) {
    if (!ctx || !d_values || !d_sigmas || !beta || !gamma || !d_z || (blinding_factors && !d_blinding)) { set_error("zkhip_permutation_products_device: null argument"); return ZKHIP_EINVAL; }
    if (ncols == 0) return ZKHIP_OK;
    if (chunk_len == 0 || k == 0 || k > 26) { set_error("zkhip_permutation_products_device: bad chunk_len / k"); return ZKHIP_EINVAL; }
    size_t n = (size_t)1 << k;
    if (blinding_factors + 1 >= n) { set_error("zkhip_permutation_products_device: too many blinding factors"); return ZKHIP_EINVAL; }
    uint32_t nsets = (uint32_t)((ncols + chunk_len - 1) / chunk_len);
    hipStream_t st = ctx->stream;
    // omega_k and its split power tables
    el2<Fr> w = from_canonical_words<Fr>(FR_ROOT_OF_UNITY);
    for (uint32_t i = k; i < FR_S; ++i) w = sqr(w);
    uint64_t w_abi[4];
    mem_store(w_abi, to_abi(w));
    const zkhip_ctx::Twiddle* tw;
    ZK_TRY(ctx->get_twiddles(w_abi, k, &tw));
    // per-column constants delta^j * beta (raw R' form)
    el2<Fr> b = from_abi<Fr>(mem_load(beta)), delta = from_canonical_words<Fr>(FR_DELTA), dj = one<Fr>();
    std::vector<fe32> dbeta(ncols);
    for (size_t j = 0; j < ncols; ++j) { dbeta[j] = fe_pack(fe_canonical<Fr>((dj * b).v)); dj = dj * delta; }
    void *d_ptrs, *d_dbeta, *d_terms;
    ZK_TRY(ctx->get_scratch("po_perm_ptrs", 2 * ncols * sizeof(void*), &d_ptrs));
    ZK_TRY(ctx->get_scratch("po_perm_dbeta", ncols * 32, &d_dbeta));
    ZK_TRY(ctx->get_scratch("po_terms", (size_t)nsets * n * 32, &d_terms));
    std::vector<const void*> ptrs(2 * ncols);
    for (size_t j = 0; j < ncols; ++j) { ptrs[j] = d_values[j]; ptrs[ncols + j] = d_sigmas[j]; }
    ZK_TRY(ctx->upload(d_ptrs, ptrs.data(), 2 * ncols * sizeof(void*)));
    ZK_TRY(ctx->upload(d_dbeta, dbeta.data(), ncols * 32));
    PermParams P;
    P.values = (const uint32_t* const*)d_ptrs;
    P.sigmas = (const uint32_t* const*)((const void**)d_ptrs + ncols);
    P.ncols = (uint32_t)ncols; P.chunk_len = chunk_len;
    P.beta = abi_to_fe(beta); P.gamma = abi_to_fe(gamma);
    P.dbeta = (const uint32_t*)d_dbeta;
    P.w_lo = (const uint32_t*)tw->d_lo; P.w_hi = (const uint32_t*)tw->d_hi; P.w_h = tw->h;
    ProfScope ps(ctx, "grand_product");
    dim3 grid(div_up(n, 256), nsets);
    hipLaunchKernelGGL(k_perm_terms<1>, grid, dim3(256), 0, st, P, n, (uint32_t*)d_terms);
    ZK_TRY(launch_batch_invert<0>(ctx, d_terms, (size_t)nsets * n));
    hipLaunchKernelGGL(k_perm_terms<0>, grid, dim3(256), 0, st, P, n, (uint32_t*)d_terms);
    return running_products(ctx, d_terms, nsets, n, n - blinding_factors, nsets, n - blinding_factors - 1, d_blinding, d_z);
}

int zkhip_lookup_product_device(zkhip_ctx* ctx, uint32_t k, const void* d_compressed_input, const void* d_compressed_table,
                                const void* d_permuted_input, const void* d_permuted_table, const uint64_t beta[4], const uint64_t gamma[4],
                                uint32_t blinding_factors, const void* d_blinding, void* d_z) {
    if (!ctx || !d_compressed_input || !d_compressed_table || !d_permuted_input || !d_permuted_table || !beta || !gamma || !d_z ||
        (blinding_factors && !d_blinding)) { set_error("zkhip_lookup_product_device: null argument"); return ZKHIP_EINVAL; }
    if (k == 0 || k > 26) { set_error("zkhip_lookup_product_device: bad k"); return ZKHIP_EINVAL; }
    size_t n = (size_t)1 << k;
    if (blinding_factors + 1 >= n) { set_error("zkhip_lookup_product_device: too many blinding factors"); return ZKHIP_EINVAL; }
    hipStream_t st = ctx->stream;
    void* d_terms;
    ZK_TRY(ctx->get_scratch("po_terms", n * 32, &d_terms));
    fe bv = abi_to_fe(beta), gv = abi_to_fe(gamma);
    ProfScope ps(ctx, "grand_product");
    hipLaunchKernelGGL(k_lookup_terms<1>, dim3(div_up(n, 256)), dim3(256), 0, st, (const uint32_t*)d_permuted_input, (const uint32_t*)d_permuted_table,
                       bv, gv, n, (uint32_t*)d_terms);
    ZK_TRY(launch_batch_invert<0>(ctx, d_terms, n));
    hipLaunchKernelGGL(k_lookup_terms<0>, dim3(div_up(n, 256)), dim3(256), 0, st, (const uint32_t*)d_compressed_input,
                       (const uint32_t*)d_compressed_table, bv, gv, n, (uint32_t*)d_terms);
    void* zs[1] = {d_z};
    return running_products(ctx, d_terms, 1, n, n - blinding_factors, 0, 0, d_blinding, zs);
}

// permutation::commit and every lookup's commit_product behind ONE batch inversion (each inversion pass has a
// ~0.3 ms latency floor: 380 dependent products on one wave).  Segments: the permutation sets, then the lookups.
// Rows [row0, row0 + count) only (count < n: a context with a communicator, every rank its own range — the terms are pointwise in the
// rows, the running products are completed across the ranks by running_products' exchange); z columns are written in that range.
static int grand_products_rows(zkhip_ctx* ctx, uint32_t k, const uint64_t beta[4], const uint64_t gamma[4], uint32_t blinding_factors,
                               const void* const* d_values, const void* const* d_sigmas, size_t ncols, uint32_t chunk_len,
                               const void* d_perm_blinding, void* const* d_perm_z,
                               size_t n_lookups, const void* const* d_compressed_input, const void* const* d_compressed_table,
                               const void* const* d_permuted_input, const void* const* d_permuted_table, const void* d_lookup_blinding,
                               void* const* d_lookup_z, size_t row0, size_t count) {
    if (!ctx || !beta || !gamma) { set_error("zkhip_grand_products_device: null argument"); return ZKHIP_EINVAL; }
    if (ncols && (!d_values || !d_sigmas || !d_perm_z || chunk_len == 0)) { set_error("zkhip_grand_products_device: bad permutation arguments"); return ZKHIP_EINVAL; }
    if (n_lookups && (!d_compressed_input || !d_compressed_table || !d_permuted_input || !d_permuted_table || !d_lookup_z)) { set_error("zkhip_grand_products_device: bad lookup arguments"); return ZKHIP_EINVAL; }
    if (k == 0 || k > 26) { set_error("zkhip_grand_products_device: bad k"); return ZKHIP_EINVAL; }
    const size_t n_total = (size_t)1 << k;
    const bool ranged = count != n_total;
    if (row0 + count > n_total || count == 0) { set_error("zkhip_grand_products_device: rows [%zu, %zu) of %zu", row0, row0 + count, n_total); return ZKHIP_EINVAL; }
    const size_t n = count, off = row0 * 32;
    const uint32_t bf = blinding_factors;
    if (bf + 1 >= n) { set_error("zkhip_grand_products_device: too many blinding factors"); return ZKHIP_EINVAL; }
    uint32_t nsets = ncols ? (uint32_t)((ncols + chunk_len - 1) / chunk_len) : 0;
    uint32_t nseg = nsets + (uint32_t)n_lookups;
    if (nseg == 0) return ZKHIP_OK;
    if (bf && ((nsets && !d_perm_blinding) || (n_lookups && !d_lookup_blinding))) { set_error("zkhip_grand_products_device: blinding rows missing"); return ZKHIP_EINVAL; }
    hipStream_t st = ctx->stream;
    void *d_terms, *d_blind;
    ZK_TRY(ctx->get_scratch("po_terms", (size_t)nseg * n * 32, &d_terms));
    ZK_TRY(ctx->get_scratch("po_blind", (size_t)nseg * (bf ? bf : 1) * 32, &d_blind));
    if (bf && nsets) ZK_HIP(hipMemcpyAsync(d_blind, d_perm_blinding, (size_t)nsets * bf * 32, hipMemcpyDeviceToDevice, st));
    if (bf && n_lookups) ZK_HIP(hipMemcpyAsync((char*)d_blind + (size_t)nsets * bf * 32, d_lookup_blinding, n_lookups * bf * 32, hipMemcpyDeviceToDevice, st));
    fe bv = abi_to_fe(beta), gv = abi_to_fe(gamma);
    PermParams P;
    memset(&P, 0, sizeof P);
    ProfScope ps(ctx, "grand_product");
    if (nsets) {
        el2<Fr> w = from_canonical_words<Fr>(FR_ROOT_OF_UNITY);
        for (uint32_t i = k; i < FR_S; ++i) w = sqr(w);
        uint64_t w_abi[4];
        mem_store(w_abi, to_abi(w));
        const zkhip_ctx::Twiddle* tw;
        ZK_TRY(ctx->get_twiddles(w_abi, k, &tw));
        // the identity permutation's value at GLOBAL row row0 + i is delta^j w^row0 w^i: the range's offset goes into the per-column constant
        el2<Fr> b = from_abi<Fr>(mem_load(beta)) * pow_u64<Fr>(w, (uint64_t)row0), delta = from_canonical_words<Fr>(FR_DELTA), dj = one<Fr>();
        std::vector<fe32> dbeta(ncols);
        for (size_t j = 0; j < ncols; ++j) { dbeta[j] = fe_pack(fe_canonical<Fr>((dj * b).v)); dj = dj * delta; }
        void *d_ptrs, *d_dbeta;
        ZK_TRY(ctx->get_scratch("po_perm_ptrs", 2 * ncols * sizeof(void*), &d_ptrs));
        ZK_TRY(ctx->get_scratch("po_perm_dbeta", ncols * 32, &d_dbeta));
        std::vector<const void*> ptrs(2 * ncols);
        for (size_t j = 0; j < ncols; ++j) { ptrs[j] = (const char*)d_values[j] + off; ptrs[ncols + j] = (const char*)d_sigmas[j] + off; }
        ZK_TRY(ctx->upload(d_ptrs, ptrs.data(), 2 * ncols * sizeof(void*)));
        ZK_TRY(ctx->upload(d_dbeta, dbeta.data(), ncols * 32));
        P.values = (const uint32_t* const*)d_ptrs;
        P.sigmas = (const uint32_t* const*)((const void**)d_ptrs + ncols);
        P.ncols = (uint32_t)ncols; P.chunk_len = chunk_len;
        P.beta = bv; P.gamma = gv;
        P.dbeta = (const uint32_t*)d_dbeta;
        P.w_lo = (const uint32_t*)tw->d_lo; P.w_hi = (const uint32_t*)tw->d_hi; P.w_h = tw->h;
        hipLaunchKernelGGL(k_perm_terms<1>, dim3(div_up(n, 256), nsets), dim3(256), 0, st, P, n, (uint32_t*)d_terms);
    }
    auto at = [&](const void* p_) { return (const uint32_t*)((const char*)p_ + off); };
    for (size_t l = 0; l < n_lookups; ++l)
        hipLaunchKernelGGL(k_lookup_terms<1>, dim3(div_up(n, 256)), dim3(256), 0, st, at(d_permuted_input[l]), at(d_permuted_table[l]), bv, gv, n,
                           (uint32_t*)d_terms + ((size_t)nsets + l) * n * 8);
    ZK_TRY(launch_batch_invert<0>(ctx, d_terms, (size_t)nseg * n));
    if (nsets) hipLaunchKernelGGL(k_perm_terms<0>, dim3(div_up(n, 256), nsets), dim3(256), 0, st, P, n, (uint32_t*)d_terms);
    for (size_t l = 0; l < n_lookups; ++l)
        hipLaunchKernelGGL(k_lookup_terms<0>, dim3(div_up(n, 256)), dim3(256), 0, st, at(d_compressed_input[l]), at(d_compressed_table[l]), bv, gv, n,
                           (uint32_t*)d_terms + ((size_t)nsets + l) * n * 8);
    std::vector<void*> zs(nseg);
    for (uint32_t i = 0; i < nsets; ++i) zs[i] = (char*)d_perm_z[i] + off;
    for (size_t l = 0; l < n_lookups; ++l) zs[nsets + l] = (char*)d_lookup_z[l] + off;
    RpShard sh{row0, n_total};
    return running_products(ctx, d_terms, nseg, n, n_total - bf, nsets, n_total - bf - 1, d_blind, zs.data(), ranged ? &sh : nullptr);
}
int zkhip_grand_products_device(zkhip_ctx* ctx, uint32_t k, const uint64_t beta[4], const uint64_t gamma[4], uint32_t blinding_factors,
                                const void* const* d_values, const void* const* d_sigmas, size_t ncols, uint32_t chunk_len,
                                const void* d_perm_blinding, void* const* d_perm_z,
                                size_t n_lookups, const void* const* d_compressed_input, const void* const* d_compressed_table,
                                const void* const* d_permuted_input, const void* const* d_permuted_table, const void* d_lookup_blinding,
                                void* const* d_lookup_z) {
    if (k == 0 || k > 26) { set_error("zkhip_grand_products_device: bad k"); return ZKHIP_EINVAL; }
    return grand_products_rows(ctx, k, beta, gamma, blinding_factors, d_values, d_sigmas, ncols, chunk_len, d_perm_blinding, d_perm_z, n_lookups,
                               d_compressed_input, d_compressed_table, d_permuted_input, d_permuted_table, d_lookup_blinding, d_lookup_z, 0,
                               (size_t)1 << k);
}

}  // extern "C"

namespace zk {
int grand_products_range(zkhip_ctx* ctx, uint32_t k, const uint64_t beta[4], const uint64_t gamma[4], uint32_t blinding_factors,
                         const void* const* d_values, const void* const* d_sigmas, size_t ncols, uint32_t chunk_len, const void* d_perm_blinding,
                         void* const* d_perm_z, size_t n_lookups, const void* const* d_compressed_input, const void* const* d_compressed_table,
                         const void* const* d_permuted_input, const void* const* d_permuted_table, const void* d_lookup_blinding,
                         void* const* d_lookup_z, size_t row0, size_t count) {
    return grand_products_rows(ctx, k, beta, gamma, blinding_factors, d_values, d_sigmas, ncols, chunk_len, d_perm_blinding, d_perm_z, n_lookups,
                               d_compressed_input, d_compressed_table, d_permuted_input, d_permuted_table, d_lookup_blinding, d_lookup_z, row0, count);
}
}  // namespace zk
