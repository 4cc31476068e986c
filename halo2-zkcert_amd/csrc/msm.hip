// msm.hip — BN254 G1 multi-scalar multiplication for gfx950.
//
// Drop-in for halo2curves 0.4.0 msm::best_multiexp as called by ParamsKZG::commit /
// commit_lagrange [UPSTREAM-RECALL; halo2curves pinned at /root/reference/Cargo.lock:1359-1361,
// reached from gen_snark_shplonk at /root/reference/src/helpers.rs:233,299].  The sum is unique,
// so the algorithm is free: this is NOT the reference's per-thread-chunk Pippenger.
//
// MI355X design:
//  * The SRS is fixed for the life of the process, so zkhip_srs_load stores, for every base P_i,
//    its W window multiples 2^(c*w) P_i in affine form (W*n*64 B; 288 GB HBM makes this cheap).
//    An MSM is then ONE bucket pass over n*W (digit, point) pairs: no per-window doublings and
//    one 2^(c-1)-bucket reduction instead of W of them.
//  * Signed c-bit digits; pairs are grouped by |digit| with an LDS-histogram radix partition, so work is
//    proportional to non-zero digits — zero / small witness values cost nothing in the upper windows.
//  * Bucket sums are segmented: every thread adds at most SEG pairs (XYZZ + affine mixed additions, 8M + 2S),
//    partial sums are folded in further rounds.  Skewed buckets (boolean columns) therefore cost
//    depth O(log), not O(count).
//  * Integer-multiply bound: 8 * 171 + 2 * 126 = 1620 v_mad_u64_u32 per (pair, window).
//  * Host slices (zkhip_msm_g1 / zkhip_msm_g1_batch, round 6): msm_local = msm_partials (everything up to the buckets' partial sums) + msm_tail (the bucket reduction),
//    so that a large host slice can be uploaded in chunks by a worker thread and accumulated chunk by chunk as the bytes land, the chunks' buckets merged
//    (k_merge_buckets) before ONE tail — see host_msm_* near the end of the file.
#include <algorithm>
#include <mutex>

#include "common.hpp"
using namespace zk;

struct zkhip_srs {
    size_t n = 0;             // bases held here (the table's row length)
    size_t first0 = 0;        // global index of base 0 of this handle (a point-range shard of a larger SRS; 0 otherwise)
    size_t n_total = 0;       // length of the whole SRS
    uint32_t c = 0, W = 0, B = 0;
    void* d_table = nullptr;  // [W][n] affine, Montgomery
};

static uint32_t pick_window(const zkhip_ctx* ctx, size_t n) {
    if (ctx->opt.msm_c >= 2 && ctx->opt.msm_c <= 20) return (uint32_t)ctx->opt.msm_c;
    uint32_t lg = 0;
    while (((size_t)1 << lg) < n) ++lg;
    // measured on MI355X (profiles/): 2^17 -> 16 (16 windows), 2^19 -> 17 (15 windows, 15 * 17 = 255 bits: no short top window
    // whose few digit values would pile a third of the points into four buckets), 2^20..2^21 -> 19 (14 windows), >= 2^22 -> 20
    // (13 windows and a 14-bit top window: -6 % against 19, whose 7-bit top window fills 64 buckets with 1/14 of all pairs and
    // forces reduction rounds; the sort's passes hold 2^8 partitions x 2^11 bins, so 20 is the largest supported)
    int c = (int)lg - 1;
    if (c < 3) c = 3;
    if (c > 16) c = lg >= 22 ? 20 : (lg >= 20 ? 19 : 17);
    return (uint32_t)c;
}

// ------------------------------------------------------------------ SRS precomputation
// Tables and scratch hold coordinates in the library's own R' = 2^261 Montgomery form, canonical,
// 32 B each ("raw"): loading one is a limb split, no multiplication.
__global__ void k_import_bases(const uint32_t* in_xy_abi, uint32_t* out_xy_raw, uint32_t* out_xyz_raw, size_t n, int in_is_raw) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    g1a a = in_is_raw ? g1a_load_raw(in_xy_abi + i * 16) : g1a_load_abi(in_xy_abi + i * 16);
    g1a_store_raw(out_xy_raw + i * 16, a);
    g1j_store_raw(out_xyz_raw + i * 24, g1j_from_affine(a));
}
// Window layout: W = ceil(255 / c) windows over the 254 bits a recoded scalar (<= (r - 1) / 2 < 2^253, plus room for the last carry)
// can occupy.  Windows 0 .. W-3 are c bits wide; the LAST TWO share the remaining R = 254 - (W - 2) c bits evenly (c = 20: 17 + 17
// instead of 20 + 14).  With a 14-bit top window every one of its n digits fell into the first 2^13 buckets — 512 entries per bucket
// at 2^22 against 96 elsewhere, i.e. 10 partial sums in those buckets and a reduction round (k_accum_jac) for every MSM; two 17-bit
// windows put 2 x 64 extra entries into the first 2^16 buckets (3.4 partial sums: none).
ZK_HD __forceinline__ void window_at(uint32_t w, uint32_t c, uint32_t W, uint32_t& off, uint32_t& bits) {
    const uint32_t base = (W - 2) * c, R = 254 - base, ca = (R + 1) / 2;
    if (w + 2 < W) { off = w * c; bits = c; }
    else if (w + 2 == W) { off = base; bits = ca; }
    else { off = base + ca; bits = R - ca; }
}
__global__ void k_pow2c(const uint32_t* in_xyz, uint32_t* out_xyz, size_t n, uint32_t c) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    g1j p = g1j_load_raw(in_xyz + i * 24);
    for (uint32_t j = 0; j < c; ++j) p = g1j_double(p);
    g1j_store_raw(out_xyz + i * 24, p);
}
// Jacobian -> affine with Montgomery's trick over G points per thread (strided for coalescing); raw in, raw out.
template <int G>
__global__ void k_batch_to_affine(const uint32_t* in_xyz, uint32_t* out_xy, size_t n) {
    size_t T = (size_t)gridDim.x * blockDim.x;
    size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    el2<Fq> prefix[G];
    el2<Fq> acc = one<Fq>();
#pragma unroll
    for (int j = 0; j < G; ++j) {
        size_t i = t + (size_t)j * T;
        el1<Fq> z = one<Fq>();
        if (i < n) {
            z = load_raw<Fq>(in_xyz + i * 24 + 16);
            if (fe_is_zero_exact(z.v)) z = one<Fq>();
        }
        prefix[j] = acc;
        acc = acc * z;
    }
    el2<Fq> inv_acc = inv<Fq>(acc);
#pragma unroll
    for (int j = G - 1; j >= 0; --j) {
        size_t i = t + (size_t)j * T;
        if (i < n) {
            g1j p = g1j_load_raw(in_xyz + i * 24);
            g1a a;
            if (g1j_is_id(p)) {
                a = g1a_identity();
            } else {
                el2<Fq> zi = inv_acc * prefix[j];
                inv_acc = inv_acc * p.z;
                auto zi2 = sqr(zi);
                a.x = p.x * zi2;
                a.y = p.y * (zi2 * zi);
            }
            g1a_store_raw(out_xy + i * 16, a);
        }
    }
}

namespace zk {
int launch_batch_to_affine(zkhip_ctx* ctx, const void* d_in_xyz, void* d_out_xy, size_t n) {
    if (n == 0) return ZKHIP_OK;
    hipLaunchKernelGGL(k_batch_to_affine<8>, dim3(div_up(div_up(n, 8), 256)), dim3(256), 0, ctx->stream, (const uint32_t*)d_in_xyz,
                       (uint32_t*)d_out_xy, n);
    ZK_LAUNCH_CHECK();
    return ZKHIP_OK;
}
}  // namespace zk

namespace zk { int srs_build_raw(zkhip_ctx* ctx, const void* d_bases_raw, size_t n, zkhip_srs** out); }
static int srs_build(zkhip_ctx* ctx, const void* d_bases, size_t n, zkhip_srs** out, int bases_are_raw) {
    if (n == 0 || n > ((size_t)1 << 26)) { set_error("zkhip_srs_load: n = %zu out of range (1..2^26)", n); return ZKHIP_EINVAL; }
    zkhip_srs* s = new zkhip_srs();
    s->n = n;
    s->n_total = n;
    s->c = pick_window(ctx, n);
    s->W = (255 + s->c - 1) / s->c;
    s->B = 1u << (s->c - 1);
    if ((size_t)s->W * n >= ((size_t)1 << 31)) { delete s; set_error("zkhip_srs_load: W*n overflows the 31-bit pair index"); return ZKHIP_EINVAL; }
    hipError_t e = zk::dev_malloc((void**)&s->d_table, (size_t)s->W * n * 64);
    if (e != hipSuccess) { (void)hipGetLastError(); delete s; set_error("hipMalloc SRS table (%zu B): %s", (size_t)s->W * n * 64, hipGetErrorString(e)); return ZKHIP_ENOMEM; }
    void *ja, *jb;
    int rc = ctx->get_scratch("srs_jac_a", n * 96, &ja);
    if (rc == ZKHIP_OK) rc = ctx->get_scratch("srs_jac_b", n * 96, &jb);
    if (rc != ZKHIP_OK) { (void)hipFree(s->d_table); delete s; return rc; }
    hipStream_t st = ctx->stream;
    unsigned g = div_up(n, 256);
    hipLaunchKernelGGL(k_import_bases, dim3(g), dim3(256), 0, st, (const uint32_t*)d_bases, (uint32_t*)s->d_table, (uint32_t*)ja, n,
                       bases_are_raw);
    constexpr int G = 8;
    unsigned gt = div_up(div_up(n, G), 256);
    for (uint32_t w = 1; w < s->W; ++w) {
        uint32_t off_prev, bits_prev;
        window_at(w - 1, s->c, s->W, off_prev, bits_prev);   // window w's multiple is 2^(bits of window w - 1) times the previous one
        hipLaunchKernelGGL(k_pow2c, dim3(g), dim3(256), 0, st, (const uint32_t*)ja, (uint32_t*)jb, n, bits_prev);
        hipLaunchKernelGGL(k_batch_to_affine<G>, dim3(gt), dim3(256), 0, st, (const uint32_t*)jb,
                           (uint32_t*)((char*)s->d_table + (size_t)w * n * 64), n);
        std::swap(ja, jb);
    }
    ZK_LAUNCH_CHECK();
    ZK_HIP(hipStreamSynchronize(st));
    *out = s;
    return ZKHIP_OK;
}

namespace zk {
int srs_build_raw(zkhip_ctx* ctx, const void* d_bases_raw, size_t n, zkhip_srs** out) { return srs_build(ctx, d_bases_raw, n, out, 1); }
}  // namespace zk

// ------------------------------------------------------------------ digit recoding + sort by bucket
// Signed digits d_w in [-(2^(c_w-1)-1), 2^(c_w-1)], sum d_w 2^(off_w) = scalar (window_at: offsets and widths).  The windows cover 254 bits so the last carry is
// zero for every canonical scalar < r < 2^254.  The n*W (digit, point) pairs are grouped by bucket |d| - 1
// with a two-level most-significant-digit radix partition whose histograms live in LDS:
//   hi pass: 256 scalars (256 W pairs) per block, P = 2^HB partitions by the top bits of the bucket;
//   lo pass: 4096-pair tiles inside one partition, 2^LB bins by the low bits; the scatter is staged through LDS so that a wave
//            stores each bin's run with consecutive lanes.
// Each pass counts, reserves contiguous space with ONE returning global atomic per (tile, bin) — a wave
// touches consecutive counters — and scatters with ranks from LDS atomics.  Per-pair global atomics and the
// 4-byte scatter over the whole n*W range, which made the first version memory-bound at 2^22, are gone.
// column pointers of a batch of <= 16 columns travel by value in the kernel arguments (dev == nullptr); longer batches through a
// device table
struct ColPtrs { const uint32_t* const* dev; const uint32_t* val[16]; };
__device__ __forceinline__ const uint32_t* col_ptr(const ColPtrs& c, uint32_t col) { return c.dev ? c.dev[col] : c.val[col]; }
struct SortGeom { uint32_t c, W, B, HB, LB, P, tile, R; };   // R: copies of the partition counters (power of two <= SORT_COPIES)   // tile: pairs of one partition handled by one workgroup of the low pass
#define SORT_TILE 4096u
#define SORT_MAXP 256u     // partitions of the high radix pass (HB <= 8)
// The high pass's 2 x P global counters per column (partition sizes, then scatter cursors) are hit by every workgroup: 16384 returning
// atomics per address at 2^22, all of them in eight 128-byte lines — measured 0.2 ms per column, half of the pass.  They are kept in
// R copies (workgroup b uses copy b mod R; k_part_scan turns the counts into per-copy bases inside each partition's range).
#define SORT_COPIES 64u
#define SORT_PSTRIDE 260u  // part_off / tile_start: P + 1 entries per column, padded

extern __shared__ uint32_t sort_lds[];   // staging area of the two scatter kernels

__device__ __forceinline__ void digit_at(const uint32_t* sl, uint32_t w, uint32_t c, uint32_t W, uint32_t& carry, uint32_t& mag, uint32_t& neg) {
    uint32_t bit, cw;
    window_at(w, c, W, bit, cw);
    const uint32_t half = 1u << (cw - 1), mask = (1u << cw) - 1;
    uint32_t limb = bit >> 5, sh = bit & 31;
    uint64_t v = 0;
    if (limb < 8) v = sl[limb] | ((uint64_t)sl[limb + 1] << 32);
    uint32_t raw = ((uint32_t)(v >> sh) & mask) + carry;
    if (raw > half) { mag = (1u << cw) - raw; neg = 1; carry = 1; } else { mag = raw; neg = 0; carry = 0; }
}

// SCATTER = false: part_cnt[p] += pairs of this block in partition p.
// SCATTER = true : tmp_entry / tmp_key get the pairs grouped by partition (part_off from k_part_scan).
template <bool SCATTER>
__global__ void __launch_bounds__(256) k_sort_hi(const ColPtrs scalar_cols, size_t n, size_t first, size_t srs_n, SortGeom g,
                                                 uint32_t* part_cnt_all, uint32_t* part_cursor_all, const uint32_t* part_off_all,
                                                 uint32_t* tmp_entry_all, uint16_t* tmp_key_all, size_t items) {
    __shared__ uint32_t sl[256][9];
    __shared__ uint32_t hist[SORT_MAXP], base[SORT_MAXP], cnt_of[SORT_MAXP];
    const uint32_t tid = threadIdx.x, col = blockIdx.y;
    const bool staged = SCATTER && g.W <= 24;   // 256 W pairs x 7 bytes of dynamic LDS
    size_t i = blockIdx.x * (size_t)256 + tid;
    const bool live = i < n;
    hist[tid] = 0;   // 256 threads, SORT_MAXP = 256
    uint32_t flip = 0;   // the scalar was replaced by r - scalar: every digit's point changes sign
    if (live) {
        fe32 sc = abi_to_canonical_words<Fr>(mem_load(col_ptr(scalar_cols, col) + (first + i) * 8));
        // Scalars above (r - 1) / 2 are recoded as -(r - s): for uniform scalars nothing changes (the digits of r - s are as dense as
        // those of s), but witness columns are full of small NEGATIVE values (-1, -x for a limb or a word x), whose canonical form has
        // every window non-zero while r - s has one or two — they become as cheap as the small positive values, whose zero digits
        // are skipped.
        constexpr uint32_t RW[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
        constexpr uint32_t HW[8] = {0xf8000000u, 0xa1f0fac9u, 0x3cdcb848u, 0x9419f424u, 0x40c0ac2eu, 0xdc2822dbu, 0x7098d014u, 0x18322739u};
        bool gt = false, decided = false;
#pragma unroll
        for (int j = 7; j >= 0; --j) {
            if (!decided && sc.w[j] != HW[j]) { gt = sc.w[j] > HW[j]; decided = true; }
        }
        if (gt) {
            uint32_t borrow = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                uint64_t d = (uint64_t)RW[j] - sc.w[j] - borrow;
                sc.w[j] = (uint32_t)d;
                borrow = (uint32_t)(d >> 63);
            }
            flip = 1;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) sl[tid][j] = sc.w[j];
        sl[tid][8] = 0;
    }
    __syncthreads();
    const uint32_t lomask = (1u << g.LB) - 1;
    if (live) {
        uint32_t carry = 0, mag, neg;
        for (uint32_t w = 0; w < g.W; ++w) {
            digit_at(sl[tid], w, g.c, g.W, carry, mag, neg);
            if (mag) atomicAdd(&hist[(mag - 1) >> g.LB], 1u);
        }
    }
    __syncthreads();
    const uint32_t copy = blockIdx.x & (g.R - 1);
    uint32_t* part_cnt = part_cnt_all + (size_t)col * SORT_MAXP * SORT_COPIES;   // [copy][partition]: counts, after k_part_scan the copy's base
    if (!SCATTER) {
        if (tid < g.P && hist[tid]) atomicAdd(&part_cnt[copy * SORT_MAXP + tid], hist[tid]);
        return;
    }
    if (tid < g.P) {
        uint32_t h = hist[tid];
        base[tid] = h ? part_off_all[(size_t)col * SORT_PSTRIDE + tid] + part_cnt[copy * SORT_MAXP + tid] +
                            atomicAdd(&part_cursor_all[((size_t)col * SORT_COPIES + copy) * SORT_MAXP + tid], h)
                      : 0u;
        cnt_of[tid] = h;
        hist[tid] = 0;
    }
    __syncthreads();
    uint32_t* tmp_entry = tmp_entry_all + (size_t)col * items;
    uint16_t* tmp_key = tmp_key_all + (size_t)col * items;
    if (!staged) {   // many windows (tiny sizes): the block's pairs do not fit in LDS, store them one by one
        if (live) {
            uint32_t carry = 0, mag, neg;
            for (uint32_t w = 0; w < g.W; ++w) {
                digit_at(sl[tid], w, g.c, g.W, carry, mag, neg);
                if (mag) {
                    uint32_t b = mag - 1, p = b >> g.LB;
                    uint32_t pos = base[p] + atomicAdd(&hist[p], 1u);
                    tmp_entry[pos] = (uint32_t)(w * srs_n + first + i) | ((neg ^ flip) << 31);
                    tmp_key[pos] = (uint16_t)(b & lomask);
                }
            }
        }
        return;
    }
    // staged: group the block's pairs by partition inside LDS, then store every partition's run with consecutive lanes
    __shared__ uint32_t lst[SORT_MAXP + 1];
    uint32_t* st_e = sort_lds;                                                // [256 W]
    uint16_t* st_k = reinterpret_cast<uint16_t*>(st_e + 256 * g.W);           // [256 W]
    uint8_t* st_p = reinterpret_cast<uint8_t*>(st_k + 256 * g.W);             // [256 W]
    {   // exclusive scan of the <= 256 partition counts: wave shuffles + the four wave totals (a serial loop on one thread cost ~2 us per block)
        __shared__ uint32_t wtot[4];
        const uint32_t lane = tid & 63, wave = tid >> 6;
        const uint32_t mine = tid < g.P ? cnt_of[tid] : 0u;
        uint32_t inc = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { uint32_t u = __shfl_up(inc, d); if ((int)lane >= d) inc += u; }
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        uint32_t before = 0;
        for (uint32_t w_ = 0; w_ < wave; ++w_) before += wtot[w_];
        if (tid < g.P) lst[tid] = before + inc - mine;
        if (tid == 255) lst[g.P] = before + inc;   // g.P <= 256: thread 255's inclusive total is the block's (counts beyond P are 0)
    }
    __syncthreads();
    if (tid < g.P) hist[tid] = lst[tid];
    __syncthreads();
    if (live) {
        uint32_t carry = 0, mag, neg;
        for (uint32_t w = 0; w < g.W; ++w) {
            digit_at(sl[tid], w, g.c, g.W, carry, mag, neg);
            if (mag) {
                uint32_t b = mag - 1, p = b >> g.LB;
                uint32_t slot = atomicAdd(&hist[p], 1u);
                st_e[slot] = (uint32_t)(w * srs_n + first + i) | ((neg ^ flip) << 31);
                st_k[slot] = (uint16_t)(b & lomask);
                st_p[slot] = (uint8_t)p;
            }
        }
    }
    __syncthreads();
    const uint32_t total = lst[g.P];
    for (uint32_t s_ = tid; s_ < total; s_ += 256) {
        uint32_t p = st_p[s_];
        uint32_t pos = base[p] + (s_ - lst[p]);
        tmp_entry[pos] = st_e[s_];
        tmp_key[pos] = st_k[s_];
    }
}

// part_off = exclusive scan of part_cnt (P + 1 entries); tile_start = exclusive scan of ceil(part_cnt / tile).
// (the R copies of a partition's count are replaced by their exclusive prefix: the copy's base inside the partition's range)
__global__ void __launch_bounds__(SORT_MAXP) k_part_scan(uint32_t* part_cnt_all, uint32_t P, uint32_t R, uint32_t tile, uint32_t* part_off_all, uint32_t* tile_start_all) {
    __shared__ uint32_t a[SORT_MAXP], b[SORT_MAXP];
    uint32_t col = blockIdx.x, t = threadIdx.x;
    uint32_t v = 0;
    if (t < P) {
        uint32_t* pc = part_cnt_all + (size_t)col * SORT_MAXP * SORT_COPIES + t;
        for (uint32_t r = 0; r < R; ++r) { uint32_t c = pc[r * SORT_MAXP]; pc[r * SORT_MAXP] = v; v += c; }
    }
    uint32_t tl = (v + tile - 1) / tile;
    a[t] = v; b[t] = tl;
    __syncthreads();
    for (uint32_t d = 1; d < SORT_MAXP; d <<= 1) {
        uint32_t x = t >= d ? a[t - d] : 0u, y = t >= d ? b[t - d] : 0u;
        __syncthreads();
        a[t] += x; b[t] += y;
        __syncthreads();
    }
    uint32_t* po = part_off_all + (size_t)col * SORT_PSTRIDE;
    uint32_t* ts = tile_start_all + (size_t)col * SORT_PSTRIDE;
    if (t < P) { po[t] = a[t] - v; ts[t] = b[t] - tl; }
    if (t == P - 1) { po[P] = a[t]; ts[P] = b[t]; }
}

// One tile (g.tile pairs) of one partition: cnt[bucket] += the tile's histogram.
__global__ void __launch_bounds__(256) k_sort_lo_count(const uint32_t* part_off_all, const uint32_t* tile_start_all, SortGeom g,
                                                       const uint16_t* tmp_key_all, size_t items, uint32_t* cnt_all) {
    __shared__ uint32_t hist[2048];
    const uint32_t tid = threadIdx.x, col = blockIdx.y, blk = blockIdx.x;
    const uint32_t* po = part_off_all + (size_t)col * SORT_PSTRIDE;
    const uint32_t* ts = tile_start_all + (size_t)col * SORT_PSTRIDE;
    if (blk >= ts[g.P]) return;
    uint32_t lo_p = 0, hi_p = g.P;   // largest p with ts[p] <= blk
    while (hi_p - lo_p > 1) {
        uint32_t mid = (lo_p + hi_p) >> 1;
        if (ts[mid] <= blk) lo_p = mid; else hi_p = mid;
    }
    const uint32_t p = lo_p;
    const uint32_t beg = po[p] + (blk - ts[p]) * g.tile, end = min(beg + g.tile, po[p + 1]);
    const uint32_t nbins = 1u << g.LB;
    for (uint32_t j = tid; j < nbins; j += 256) hist[j] = 0;
    __syncthreads();
    const uint16_t* tmp_key = tmp_key_all + (size_t)col * items;
    for (uint32_t j = beg + tid; j < end; j += 256) atomicAdd(&hist[tmp_key[j]], 1u);
    __syncthreads();
    uint32_t* cnt = cnt_all + (size_t)col * g.B;
    const uint32_t bucket0 = p << g.LB;
    for (uint32_t j = tid; j < nbins; j += 256)
        if (hist[j]) atomicAdd(&cnt[bucket0 + j], hist[j]);
}

// The final scatter, staged: a tile's pairs are first grouped by bin inside LDS (ranks from LDS atomics), then written out in
// bin order, so that the lanes of a wave store to consecutive addresses within every bin's run (the direct version issued one
// isolated 4-byte store per pair: each cost a whole memory sector).
__global__ void __launch_bounds__(256) k_sort_lo_staged(const uint32_t* part_off_all, const uint32_t* tile_start_all, SortGeom g,
                                                        const uint32_t* tmp_entry_all, const uint16_t* tmp_key_all, size_t items,
                                                        const uint32_t* off_all, uint32_t* cursor_all, uint32_t* entries_all) {
    __shared__ uint32_t w_tot[4];
    const uint32_t tid = threadIdx.x, col = blockIdx.y, blk = blockIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t* po = part_off_all + (size_t)col * SORT_PSTRIDE;
    const uint32_t* ts = tile_start_all + (size_t)col * SORT_PSTRIDE;
    if (blk >= ts[g.P]) return;
    uint32_t lo_p = 0, hi_p = g.P;   // largest p with ts[p] <= blk
    while (hi_p - lo_p > 1) {
        uint32_t mid = (lo_p + hi_p) >> 1;
        if (ts[mid] <= blk) lo_p = mid; else hi_p = mid;
    }
    const uint32_t p = lo_p;
    const uint32_t beg = po[p] + (blk - ts[p]) * g.tile, end = min(beg + g.tile, po[p + 1]), cnt = end - beg;
    const uint32_t nbins = 1u << g.LB;
    uint32_t* hist = sort_lds;                 // [nbins] counts, then running cursors
    uint32_t* base = hist + nbins;             // [nbins] global position of the bin's run
    uint32_t* lst = base + nbins;              // [nbins] start of the bin inside the staged tile
    uint32_t* st_e = lst + nbins;              // [tile] staged entries
    uint16_t* st_k = reinterpret_cast<uint16_t*>(st_e + g.tile);   // [tile] their bins
    for (uint32_t j = tid; j < nbins; j += 256) hist[j] = 0;
    __syncthreads();
    const uint16_t* tmp_key = tmp_key_all + (size_t)col * items;
    for (uint32_t j = beg + tid; j < end; j += 256) atomicAdd(&hist[tmp_key[j]], 1u);
    __syncthreads();
    // exclusive scan of the counts (8 bins per thread for 2048 bins), and one returning global atomic per non-empty bin
    const uint32_t bucket0 = p << g.LB;
    const uint32_t* off = off_all + (size_t)col * (g.B + 4);
    uint32_t* cursor = cursor_all + (size_t)col * g.B;
    const uint32_t per = (nbins + 255) / 256;   // thread t owns bins t, t + 256, ...: see k_sort_lo_staged16
    uint32_t my = 0;
    for (uint32_t q = 0; q < per; ++q) { uint32_t j = q * 256 + tid; if (j < nbins) my += hist[j]; }
    uint32_t inc = my;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { uint32_t u = __shfl_up(inc, d); if ((int)lane >= d) inc += u; }
    if (lane == 63) w_tot[wave] = inc;
    __syncthreads();
    uint32_t run = inc - my;
    for (uint32_t w = 0; w < wave; ++w) run += w_tot[w];
    for (uint32_t q = 0; q < per; ++q) {
        uint32_t j = q * 256 + tid;
        if (j < nbins) {
            uint32_t h = hist[j];
            base[j] = h ? off[bucket0 + j] + atomicAdd(&cursor[bucket0 + j], h) : 0u;
            lst[j] = run;
            hist[j] = run;
            run += h;
        }
    }
    __syncthreads();
    const uint32_t* tmp_entry = tmp_entry_all + (size_t)col * items;
    for (uint32_t j = beg + tid; j < end; j += 256) {
        uint32_t key = tmp_key[j];
        uint32_t slot = atomicAdd(&hist[key], 1u);
        st_e[slot] = tmp_entry[j];
        st_k[slot] = (uint16_t)key;
    }
    __syncthreads();
    uint32_t* entries = entries_all + (size_t)col * items;
    for (uint32_t i = tid; i < cnt; i += 256) {
        uint32_t key = st_k[i];
        entries[base[key] + (i - lst[key])] = st_e[i];
    }
}

// The same pass for the default 4096-pair tile with ONE LDS atomic per pair: the counting atomic's return value is the pair's
// rank inside its bin, kept in registers (16 pairs per thread) until the bins' starts are known.
template <int PER_T, int NT>
__global__ void __launch_bounds__(NT) k_sort_lo_staged16(const uint32_t* part_off_all, const uint32_t* tile_start_all, SortGeom g,
                                                          const uint32_t* tmp_entry_all, const uint16_t* tmp_key_all, size_t items,
                                                          const uint32_t* off_all, uint32_t* cursor_all, uint32_t* entries_all) {
    __shared__ uint32_t w_tot[NT / 64];
    const uint32_t tid = threadIdx.x, col = blockIdx.y, blk = blockIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t* po = part_off_all + (size_t)col * SORT_PSTRIDE;
    const uint32_t* ts = tile_start_all + (size_t)col * SORT_PSTRIDE;
    if (blk >= ts[g.P]) return;
    uint32_t lo_p = 0, hi_p = g.P;   // largest p with ts[p] <= blk
    while (hi_p - lo_p > 1) {
        uint32_t mid = (lo_p + hi_p) >> 1;
        if (ts[mid] <= blk) lo_p = mid; else hi_p = mid;
    }
    const uint32_t p = lo_p;
    const uint32_t beg = po[p] + (blk - ts[p]) * (PER_T * (uint32_t)NT), end = min(beg + (PER_T * (uint32_t)NT), po[p + 1]), cnt = end - beg;
    const uint32_t nbins = 1u << g.LB;
    uint32_t* hist = sort_lds;                 // [nbins] counts
    uint32_t* base = hist + nbins;             // [nbins] global position of the bin's run
    uint32_t* lst = base + nbins;              // [nbins] start of the bin inside the staged tile
    uint32_t* st_e = lst + nbins;              // [tile] staged entries
    uint16_t* st_k = reinterpret_cast<uint16_t*>(st_e + (PER_T * (uint32_t)NT));   // [tile] their bins
    for (uint32_t j = tid; j < nbins; j += NT) hist[j] = 0;
    __syncthreads();
    const uint16_t* tmp_key = tmp_key_all + (size_t)col * items;
    uint32_t keys[PER_T], ranks[PER_T];
#pragma unroll
    for (int q = 0; q < PER_T; ++q) {
        uint32_t j = beg + tid + q * NT;
        keys[q] = j < end ? tmp_key[j] : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int q = 0; q < PER_T; ++q) ranks[q] = keys[q] != 0xFFFFFFFFu ? atomicAdd(&hist[keys[q]], 1u) : 0u;
    __syncthreads();
    const uint32_t bucket0 = p << g.LB;
    const uint32_t* off = off_all + (size_t)col * (g.B + 4);
    uint32_t* cursor = cursor_all + (size_t)col * g.B;
    // Thread t owns bins t, t + NT, ... (NOT a contiguous range): the order of the bins' runs inside the staged tile is irrelevant — the
    // write-out addresses every pair through base[] / lst[] of its own bin — and with this ownership the lanes of a wave read
    // consecutive LDS words (a contiguous range per thread is an 8-way / 2-way bank conflict on three arrays) and issue their returning
    // global atomics on consecutive cursor words, i.e. one cache line per wave instead of one per lane.
    const uint32_t per = (nbins + NT - 1) / NT;
    uint32_t my = 0;
    for (uint32_t q = 0; q < per; ++q) { uint32_t j = q * NT + tid; if (j < nbins) my += hist[j]; }
    uint32_t inc = my;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { uint32_t u = __shfl_up(inc, d); if ((int)lane >= d) inc += u; }
    if (lane == 63) w_tot[wave] = inc;
    __syncthreads();
    uint32_t run = inc - my;
    for (uint32_t w = 0; w < wave; ++w) run += w_tot[w];
    for (uint32_t q = 0; q < per; ++q) {
        uint32_t j = q * NT + tid;
        if (j < nbins) {
            uint32_t h = hist[j];
            base[j] = h ? off[bucket0 + j] + atomicAdd(&cursor[bucket0 + j], h) : 0u;
            lst[j] = run;
            run += h;
        }
    }
    __syncthreads();
    const uint32_t* tmp_entry = tmp_entry_all + (size_t)col * items;
#pragma unroll
    for (int q = 0; q < PER_T; ++q) {
        if (keys[q] != 0xFFFFFFFFu) {
            uint32_t slot = lst[keys[q]] + ranks[q];
            st_e[slot] = tmp_entry[beg + tid + q * NT];
            st_k[slot] = (uint16_t)keys[q];
        }
    }
    __syncthreads();
    uint32_t* entries = entries_all + (size_t)col * items;
    for (uint32_t i = tid; i < cnt; i += NT) {
        uint32_t key = st_k[i];
        entries[base[key] + (i - lst[key])] = st_e[i];
    }
}

// ceil(v / seg) without a hardware divide: seg_magic = ceil(2^32 / seg); exact for v * seg < 2^32 (v < 2^26, seg <= 64).
__device__ __forceinline__ uint32_t ceil_div_magic(uint32_t v, uint32_t seg, uint32_t seg_magic) {
    return seg == 1 ? v : __umulhi(v + seg - 1, seg_magic);
}
// Scans over the per-bucket counters, spread over B / 2048 blocks per column (one CU moves only ~10 B/clk):
//   k_plan_sums : per block, sum of cnt_in, sum of ceil(cnt_in / seg), max of cnt_in
//   k_plan_apply: every block adds up the sums of the blocks before it, then scans its own 2048 counters:
//                 off_in (if non-null) = exclusive scan of cnt_in; cnt_out = ceil(cnt_in / seg);
//                 off_out = exclusive scan of cnt_out (B + 1 entries each); max_out[col] = max cnt_in.
#define PLAN_PER 8
#define PLAN_BLOCK (256 * PLAN_PER)
// lane mode (lane_off != null): the scanned quantity is not cnt_in[b] but the number of round-0 lanes of width L touching bucket b,
// floor((off + cnt - 1) / L) - floor(off / L) + 1, from the bucket offsets of the previous scan
// Round-0 lane width actually used for a column with `total` non-zero digits out of `items` slots.  The host picks L for dense
// scalars (about four waves per SIMD over the launch); columns of small values (lookup inputs, bit columns: most digits zero)
// would leave a few hundred lanes running L dependent additions each, so the width shrinks with the column's real entry count —
// never below what keeps the lanes within the launched grid and the partial sums within their buffer (sized for L >= 8).
__device__ __forceinline__ uint32_t lane_len(uint32_t total, uint32_t items, uint32_t L, uint32_t ncols) {
    uint64_t a = ((uint64_t)total * L + items - 1) / items;
    uint64_t b = ((uint64_t)total * ncols + 262143) / 262144;
    uint32_t l = (uint32_t)(a > b ? a : b);
    return min(L, max(2u, l));
}
__device__ __forceinline__ void plan_load(const uint32_t* cnt_in, uint32_t B, uint32_t lo, uint32_t v[PLAN_PER], const uint32_t* lane_off = nullptr,
                                          uint32_t lane_L = 0, uint32_t lane_items = 0) {
    if (lane_off) {
        if (lane_items) lane_L = lane_len(lane_off[B], lane_items, lane_L, gridDim.y);
#pragma unroll
        for (int q = 0; q < PLAN_PER; ++q) {
            uint32_t c = lo + q < B ? cnt_in[lo + q] : 0u;
            uint32_t o = lo + q < B ? lane_off[lo + q] : 0u;
            v[q] = c ? (o + c - 1) / lane_L - o / lane_L + 1 : 0u;
        }
        return;
    }
    if (lo + PLAN_PER <= B) {
#pragma unroll
        for (int q = 0; q < PLAN_PER / 4; ++q) {
            uint4 x = reinterpret_cast<const uint4*>(cnt_in + lo)[q];
            v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
        }
    } else {
#pragma unroll
        for (int q = 0; q < PLAN_PER; ++q) v[q] = lo + q < B ? cnt_in[lo + q] : 0u;
    }
}
__global__ void __launch_bounds__(256) k_plan_sums(const uint32_t* cnt_in_all, uint32_t B, uint32_t seg, uint32_t seg_magic,
                                                   uint32_t* sums_all /* [col][nblk][4] */, const uint32_t* lane_off_all, uint32_t lane_L,
                                                   uint32_t lane_items) {
    __shared__ uint32_t w_a[4], w_b[4], w_m[4];
    uint32_t col = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    uint32_t v[PLAN_PER];
    plan_load(cnt_in_all + (size_t)col * B, B, blockIdx.x * PLAN_BLOCK + t * PLAN_PER, v,
              lane_off_all ? lane_off_all + (size_t)col * (B + 4) : nullptr, lane_L, lane_items);
    uint32_t sa = 0, sb = 0, m = 0;
#pragma unroll
    for (int q = 0; q < PLAN_PER; ++q) { sa += v[q]; sb += ceil_div_magic(v[q], seg, seg_magic); m = max(m, v[q]); }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { sa += __shfl_xor(sa, d); sb += __shfl_xor(sb, d); m = max(m, __shfl_xor(m, d)); }
    if (lane == 0) { w_a[wave] = sa; w_b[wave] = sb; w_m[wave] = m; }
    __syncthreads();
    if (t == 0) {
        uint32_t* o = sums_all + ((size_t)col * gridDim.x + blockIdx.x) * 4;
        o[0] = w_a[0] + w_a[1] + w_a[2] + w_a[3];
        o[1] = w_b[0] + w_b[1] + w_b[2] + w_b[3];
        o[2] = max(max(w_m[0], w_m[1]), max(w_m[2], w_m[3]));
    }
}
__global__ void __launch_bounds__(256) k_plan_apply(const uint32_t* cnt_in_all, uint32_t B, uint32_t seg, uint32_t seg_magic,
                                                    const uint32_t* sums_all, uint32_t* off_in_all, uint32_t* cnt_out_all,
                                                    uint32_t* off_out_all, uint32_t* max_out, const uint32_t* lane_off_all, uint32_t lane_L,
                                                    uint32_t max_tag, uint32_t lane_items) {
    __shared__ uint32_t w_a[4], w_b[4], s_base[3];
    uint32_t col = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6, nblk = gridDim.x;
    const uint32_t* sums = sums_all + (size_t)col * nblk * 4;
    // base of this block = sums of the blocks before it; the last block also publishes the totals
    if (wave == 0) {
        uint32_t ba = 0, bb = 0, bm = 0, ta = 0, tb = 0;
        for (uint32_t j = lane; j < nblk; j += 64) {
            uint32_t a = sums[j * 4], b = sums[j * 4 + 1];
            if (j < blockIdx.x) { ba += a; bb += b; }
            ta += a; tb += b; bm = max(bm, sums[j * 4 + 2]);
        }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            ba += __shfl_xor(ba, d); bb += __shfl_xor(bb, d); ta += __shfl_xor(ta, d); tb += __shfl_xor(tb, d);
            bm = max(bm, __shfl_xor(bm, d));
        }
        if (lane == 0) {
            s_base[0] = ba; s_base[1] = bb;
            if (blockIdx.x == nblk - 1) {
                if (off_in_all) off_in_all[(size_t)col * (B + 4) + B] = ta;
                if (off_out_all) off_out_all[(size_t)col * (B + 4) + B] = tb;
                // max_tag != 0: max_out is pinned host memory the host polls — one 8-byte system-scope store carries the value and
                // the tag of this call, so the host needs neither an event nor the end of the kernel to read it
                if (max_out && max_tag)
                    __hip_atomic_store(reinterpret_cast<uint64_t*>(max_out) + col, ((uint64_t)max_tag << 32) | bm, __ATOMIC_RELEASE,
                                       __HIP_MEMORY_SCOPE_SYSTEM);
                else if (max_out) max_out[col] = bm;
            }
        }
    }
    uint32_t lo = blockIdx.x * PLAN_BLOCK + t * PLAN_PER;
    uint32_t v[PLAN_PER], sv[PLAN_PER];
    plan_load(cnt_in_all + (size_t)col * B, B, lo, v, lane_off_all ? lane_off_all + (size_t)col * (B + 4) : nullptr, lane_L, lane_items);
    uint32_t sa = 0, sb = 0;
#pragma unroll
    for (int q = 0; q < PLAN_PER; ++q) { sv[q] = ceil_div_magic(v[q], seg, seg_magic); sa += v[q]; sb += sv[q]; }
    uint32_t ia = sa, ib = sb;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t ua = __shfl_up(ia, d), ub = __shfl_up(ib, d);
        if ((int)lane >= d) { ia += ua; ib += ub; }
    }
    if (lane == 63) { w_a[wave] = ia; w_b[wave] = ib; }
    __syncthreads();
    uint32_t ra = s_base[0] + ia - sa, rb = s_base[1] + ib - sb;
    for (uint32_t w = 0; w < wave; ++w) { ra += w_a[w]; rb += w_b[w]; }
    uint32_t* off_in = off_in_all ? off_in_all + (size_t)col * (B + 4) : nullptr;
    uint32_t* off_out = off_out_all ? off_out_all + (size_t)col * (B + 4) : nullptr;
    uint32_t* cnt_out = cnt_out_all ? cnt_out_all + (size_t)col * B : nullptr;
    uint32_t oa[PLAN_PER], ob[PLAN_PER];
#pragma unroll
    for (int q = 0; q < PLAN_PER; ++q) { oa[q] = ra; ob[q] = rb; ra += v[q]; rb += sv[q]; }
    if (lo + PLAN_PER <= B) {
#pragma unroll
        for (int q = 0; q < PLAN_PER / 4; ++q) {
            if (off_in) reinterpret_cast<uint4*>(off_in + lo)[q] = make_uint4(oa[4 * q], oa[4 * q + 1], oa[4 * q + 2], oa[4 * q + 3]);
            if (off_out) reinterpret_cast<uint4*>(off_out + lo)[q] = make_uint4(ob[4 * q], ob[4 * q + 1], ob[4 * q + 2], ob[4 * q + 3]);
            if (cnt_out) reinterpret_cast<uint4*>(cnt_out + lo)[q] = make_uint4(sv[4 * q], sv[4 * q + 1], sv[4 * q + 2], sv[4 * q + 3]);
        }
    } else {
#pragma unroll
        for (int q = 0; q < PLAN_PER; ++q) {
            if (lo + q < B) {
                if (off_in) off_in[lo + q] = oa[q];
                if (off_out) off_out[lo + q] = ob[q];
                if (cnt_out) cnt_out[lo + q] = sv[q];
            }
        }
    }
}
static int launch_plan(zkhip_ctx* ctx, unsigned ncols, const uint32_t* cnt_in, uint32_t B, uint32_t seg, uint32_t* off_in,
                       uint32_t* cnt_out, uint32_t* off_out, uint32_t* max_out, const uint32_t* lane_off = nullptr, uint32_t lane_L = 0,
                       uint32_t max_tag = 0, uint32_t lane_items = 0) {
    uint32_t magic = seg > 1 ? (uint32_t)((((uint64_t)1 << 32) + seg - 1) / seg) : 0;
    unsigned nblk = div_up(B, PLAN_BLOCK);
    void* d_sums;
    ZK_TRY(ctx->get_scratch("msm_plan_sums", (size_t)ncols * nblk * 16, &d_sums));
    hipLaunchKernelGGL(k_plan_sums, dim3(nblk, ncols), dim3(256), 0, ctx->stream, cnt_in, B, seg, magic, (uint32_t*)d_sums, lane_off, lane_L, lane_items);
    hipLaunchKernelGGL(k_plan_apply, dim3(nblk, ncols), dim3(256), 0, ctx->stream, cnt_in, B, seg, magic, (const uint32_t*)d_sums, off_in,
                       cnt_out, off_out, max_out, lane_off, lane_L, max_tag, lane_items);
    return ZKHIP_OK;
}

__device__ __forceinline__ uint32_t find_bucket(const uint32_t* off, uint32_t B, uint32_t t) {
    // largest b in [0, B) with off[b] <= t, given off[B] > t
    uint32_t lo = 0, hi = B;
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (off[mid] <= t) lo = mid; else hi = mid;
    }
    return lo;
}

// Partial sums travel between the MSM kernels as the accumulator's 4 x 9 limbs, unreduced (144 B): writing one is 9 stores and
// no arithmetic, which matters in round 0 where a lane flushes in the middle of its run while the rest of the wave waits.
#define PART_WORDS 36
__device__ __forceinline__ void g1x_store_loose(uint32_t* p, const g1x& v) {
    const fe* c[4] = {&v.x.v, &v.y.v, &v.zz.v, &v.zzz.v};
    uint32_t w[PART_WORDS];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 9; ++i) w[k * 9 + i] = c[k]->l[i];
#pragma unroll
    for (int q = 0; q < PART_WORDS / 4; ++q) reinterpret_cast<uint4*>(p)[q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
}
__device__ __forceinline__ g1x g1x_load_loose(const uint32_t* p) {
    uint32_t w[PART_WORDS];
#pragma unroll
    for (int q = 0; q < PART_WORDS / 4; ++q) {
        uint4 x = reinterpret_cast<const uint4*>(p)[q];
        w[4 * q] = x.x; w[4 * q + 1] = x.y; w[4 * q + 2] = x.z; w[4 * q + 3] = x.w;
    }
    g1x v;
    fe* c[4] = {&v.x.v, &v.y.v, &v.zz.v, &v.zzz.v};
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 9; ++i) c[k]->l[i] = w[k * 9 + i];
    return v;
}

// Round 0, lane-balanced: lane t sums the L consecutive sorted entries [t L, (t+1) L) whatever buckets they belong to, and
// writes one partial per bucket it touches (partial t - floor(off[b] / L) of bucket b).  Every lane of a wave runs the same
// trip count, where per-bucket segments left ~20 % of the lanes idle behind the longest segment.
__global__ void __launch_bounds__(256) k_accum_affine(const ColPtrs tables, const uint32_t* entries_all, size_t items,
                                                      const uint32_t* off_all, const uint32_t* segoff_all, uint32_t B, uint32_t L,
                                                      uint32_t* partial_all, size_t partial_stride, uint32_t adaptive) {
    uint32_t col = blockIdx.y;
    const uint32_t* off = off_all + (size_t)col * (B + 4);
    const uint32_t* segoff = segoff_all + (size_t)col * (B + 4);
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t total = off[B];
    if (adaptive) L = lane_len(total, (uint32_t)items, L, gridDim.y);
    if ((uint64_t)t * L >= total) return;
    const uint32_t start = t * L, end = min(start + L, total);
    const uint32_t* entries = entries_all + (size_t)col * items;
    const uint32_t* table = col_ptr(tables, col);
    uint32_t* out = partial_all + (size_t)col * partial_stride * PART_WORDS;
    uint32_t b = find_bucket(off, B, start);
    uint32_t bend = off[b + 1];
    // The lane's first entry STARTS its sum (every lane, the same iteration): until round 4 it was added to the identity — the whole
    // 8M + 2S formula with its result replaced by the fix-up — one addition in L wasted: 1.6 % of the kernel at L = 64, 6 % at 16 and
    // half of it for the two-entry lanes of sparse columns.  (A run that crosses a bucket boundary mid-lane still restarts from the identity:
    // its lane would only wait for the others' addition.)
    const uint32_t e0 = entries[start];
    g1x acc = g1x_from_affine(g1a_load_raw_cneg(table + (size_t)(e0 & 0x7fffffffu) * 16, (e0 >> 31) != 0));
    for (uint32_t j = start + 1; j < end; ++j) {
        if (j >= bend) {   // the run crosses into the next non-empty bucket
            g1x_store_loose(out + (size_t)(segoff[b] + t - off[b] / L) * PART_WORDS, acc);
            do { ++b; bend = off[b + 1]; } while (bend <= j);
            acc = g1x_identity();
        }
        uint32_t e = entries[j];
        acc = g1x_add_mixed(acc, g1a_load_raw_cneg(table + (size_t)(e & 0x7fffffffu) * 16, (e >> 31) != 0));
    }
    g1x_store_loose(out + (size_t)(segoff[b] + t - off[b] / L) * PART_WORDS, acc);
}
// Rounds >= 1: segment sums of Jacobian partials.
__global__ void __launch_bounds__(256) k_accum_jac(const uint32_t* in_all, size_t in_stride, const uint32_t* cnt_all,
                                                   const uint32_t* off_all, const uint32_t* segoff_all, uint32_t B,
                                                   uint32_t seg, uint32_t* out_all, size_t out_stride) {
    uint32_t col = blockIdx.y;
    const uint32_t* segoff = segoff_all + (size_t)col * (B + 4);
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= segoff[B]) return;
    const uint32_t* cnt = cnt_all + (size_t)col * B;
    const uint32_t* off = off_all + (size_t)col * (B + 4);
    const uint32_t* in = in_all + (size_t)col * in_stride * PART_WORDS;
    uint32_t b = find_bucket(segoff, B, t);
    uint32_t s = t - segoff[b];
    // the bucket's ceil(cnt / seg) segments are of equal length (within one): lanes of a wave then run nearly the same trip count
    const uint32_t cb = cnt[b], nsb = segoff[b + 1] - segoff[b], len = (cb + nsb - 1) / nsb;
    uint32_t lo = off[b] + s * len, hi = min(lo + len, off[b] + cb);
    g1x acc = g1x_load_loose(in + (size_t)lo * PART_WORDS);
    for (uint32_t j = lo + 1; j < hi; ++j) acc = g1x_add(acc, g1x_load_loose(in + (size_t)j * PART_WORDS));
    g1x_store_loose(out_all + ((size_t)col * out_stride + t) * PART_WORDS, acc);
}

// ---- quad-cooperative XYZZ arithmetic for the latency-bound tail -------------------------------------------------
// The bucket reduction is a chain of ~45 dependent point operations executed by a few thousand threads: the chip idles
// and the chain's length is what costs.  Here FOUR adjacent lanes hold the same operands and each computes one of the
// up-to-four independent field products of a dependency level (operands picked by lane & 3, ONE shared instruction
// stream, results exchanged with DPP quad broadcasts): an addition is 4 product levels instead of 14 products, a doubling
// 3 instead of 9.  Every lane of the quad returns the full result.
// The broadcast is written as inline assembly: with the update_dpp builtin the compiler folds the DPP move into the consuming
// subtraction and (observed on ROCm 7.2, gfx950) produces a wrong doubling.  One s_nop covers the VALU-write -> DPP-read hazard.
template <int K, int B>
__device__ __forceinline__ el<Fq, B> quad_bcast(const el<Fq, B>& a) {
    static_assert(K >= 0 && K < 4, "quad lane");
    el<Fq, B> r;
#define ZK_QB(o, i) "v_mov_b32_dpp %" #o ", %" #i " quad_perm:[%18,%18,%18,%18] row_mask:0xf bank_mask:0xf\n"
    asm volatile("s_nop 1\n" ZK_QB(0, 9) ZK_QB(1, 10) ZK_QB(2, 11) ZK_QB(3, 12) ZK_QB(4, 13) ZK_QB(5, 14) ZK_QB(6, 15) ZK_QB(7, 16) ZK_QB(8, 17)
                 : "=&v"(r.v.l[0]), "=&v"(r.v.l[1]), "=&v"(r.v.l[2]), "=&v"(r.v.l[3]), "=&v"(r.v.l[4]), "=&v"(r.v.l[5]), "=&v"(r.v.l[6]),
                   "=&v"(r.v.l[7]), "=&v"(r.v.l[8])
                 : "v"(a.v.l[0]), "v"(a.v.l[1]), "v"(a.v.l[2]), "v"(a.v.l[3]), "v"(a.v.l[4]), "v"(a.v.l[5]), "v"(a.v.l[6]), "v"(a.v.l[7]),
                   "v"(a.v.l[8]), "n"(K));
#undef ZK_QB
    return r;
}
template <int B>
__device__ __forceinline__ el<Fq, B> quad_pick(uint32_t q, const el<Fq, B>& a0, const el<Fq, B>& a1, const el<Fq, B>& a2, const el<Fq, B>& a3) {
    el<Fq, B> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        uint32_t lo = (q & 1) ? a1.v.l[i] : a0.v.l[i];
        uint32_t hi = (q & 1) ? a3.v.l[i] : a2.v.l[i];
        r.v.l[i] = (q & 2) ? hi : lo;
    }
    return r;
}
__device__ inline g1x g1x_double_q4(const g1x& p, uint32_t q) {
    if (g1x_is_id(p)) return p;
    using A1 = el<Fq, 12 * U>;
    auto u = mul_small<2>(p.y);                                   // 12 p
    // level 1: v = u^2 | xx = x^2
    A1 a = quad_pick<12 * U>(q, A1(u), A1(p.x), A1(u), A1(p.x));
    auto r1 = a * a;
    auto v = quad_bcast<0>(r1), xx = quad_bcast<1>(r1);
    auto m = mul_small<3>(xx);
    // level 2: w = u v | s = x v | mm = m^2 | zz3 = v zz
    using B2 = decltype(m);
    auto r2 = quad_pick<12 * U>(q, A1(u), A1(p.x), A1(m), A1(v)) * quad_pick(q, B2(v), B2(v), m, B2(p.zz));
    auto w = quad_bcast<0>(r2), s_ = quad_bcast<1>(r2), mm = quad_bcast<2>(r2), zz3 = quad_bcast<3>(r2);
    auto x3 = mm - mul_small<2>(s_);
    // level 3: t = m (s - x3) | wy = w y | zzz3 = w zzz
    auto d = s_ - x3;
    using B3 = decltype(d);
    using A3 = decltype(m);
    auto r3 = quad_pick(q, m, A3(w), A3(w), A3(w)) * quad_pick(q, d, B3(p.y), B3(p.zzz), B3(p.zzz));
    g1x o;
    o.x = x3;
    o.y = quad_bcast<0>(r3) - quad_bcast<1>(r3);
    o.zz = zz3;
    o.zzz = quad_bcast<2>(r3);
    return o;
}
__device__ inline g1x g1x_add_q4(const g1x& p, const g1x& g, uint32_t q) {
    if (g1x_is_id(p)) return g;
    if (g1x_is_id(g)) return p;
    // level 1: u2 = g.x p.zz | s2 = g.y p.zzz | u1 = p.x g.zz | s1 = p.y g.zzz
    using A1 = el<Fq, XBX>;
    auto r1 = quad_pick<XBX>(q, g.x, A1(g.y), p.x, A1(p.y)) * quad_pick(q, p.zz, p.zzz, g.zz, g.zzz);
    auto u2 = quad_bcast<0>(r1), s2 = quad_bcast<1>(r1), u1 = quad_bcast<2>(r1), s1 = quad_bcast<3>(r1);
    auto pp_ = u2 - u1;
    auto r = s2 - s1;
    if (is_zero(pp_)) {
        if (is_zero(r)) return g1x_double_q4(p, q);
        return g1x_identity();
    }
    // level 2: pp = pp_^2 | rr = r^2 | zz12 = p.zz g.zz | zzz12 = p.zzz g.zzz
    using A2 = decltype(pp_);
    auto r2 = quad_pick(q, pp_, r, A2(p.zz), A2(p.zzz)) * quad_pick(q, pp_, r, A2(g.zz), A2(g.zzz));
    auto pp = quad_bcast<0>(r2), rr = quad_bcast<1>(r2), zz12 = quad_bcast<2>(r2), zzz12 = quad_bcast<3>(r2);
    // level 3: ppp = pp_ pp | q_ = u1 pp | zz3 = zz12 pp
    auto r3 = quad_pick(q, pp_, A2(u1), A2(zz12), A2(zz12)) * pp;
    auto ppp = quad_bcast<0>(r3), q_ = quad_bcast<1>(r3), zz3 = quad_bcast<2>(r3);
    auto x3 = rr - (ppp + mul_small<2>(q_));
    // level 4: t = s1 ppp | zzz3 = zzz12 ppp | y' = r (q_ - x3)
    auto d = q_ - x3;
    using B4 = decltype(d);
    auto r4 = quad_pick(q, A2(s1), A2(zzz12), r, r) * quad_pick(q, B4(ppp), B4(ppp), d, d);
    g1x o;
    o.x = x3;
    o.y = quad_bcast<2>(r4) - quad_bcast<0>(r4);
    o.zz = zz3;
    o.zzz = quad_bcast<1>(r4);
    return o;
}

// Rounds >= 1 when there are too few segments to fill the chip: one quad per segment (latency-bound regime).
__global__ void __launch_bounds__(256) k_accum_jac_q4(const uint32_t* in_all, size_t in_stride, const uint32_t* cnt_all,
                                                      const uint32_t* off_all, const uint32_t* segoff_all, uint32_t B,
                                                      uint32_t seg, uint32_t* out_all, size_t out_stride) {
    uint32_t col = blockIdx.y;
    const uint32_t* segoff = segoff_all + (size_t)col * (B + 4);
    uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t t = lane >> 2, q = lane & 3;
    if (t >= segoff[B]) return;
    const uint32_t* cnt = cnt_all + (size_t)col * B;
    const uint32_t* off = off_all + (size_t)col * (B + 4);
    const uint32_t* in = in_all + (size_t)col * in_stride * PART_WORDS;
    uint32_t b = find_bucket(segoff, B, t);
    uint32_t s = t - segoff[b];
    // the bucket's ceil(cnt / seg) segments are of equal length (within one): lanes of a wave then run nearly the same trip count
    const uint32_t cb = cnt[b], nsb = segoff[b + 1] - segoff[b], len = (cb + nsb - 1) / nsb;
    uint32_t lo = off[b] + s * len, hi = min(lo + len, off[b] + cb);
    g1x acc = g1x_load_loose(in + (size_t)lo * PART_WORDS);
    for (uint32_t j = lo + 1; j < hi; ++j) acc = g1x_add_q4(acc, g1x_load_loose(in + (size_t)j * PART_WORDS), q);
    if (q == 0) g1x_store_loose(out_all + ((size_t)col * out_stride + t) * PART_WORDS, acc);
}

// The tail in two shapes: LANES = 4 (quad-cooperative; the latency regime of a few columns) and LANES = 1 (one lane per chunk; a
// wide batch has enough chunks to fill the chip, where the quads' 4x lane usage would cost throughput).
template <int LANES>
__device__ __forceinline__ g1x tail_add(const g1x& a, const g1x& b, uint32_t q) {
    if (LANES == 4) return g1x_add_q4(a, b, q);
    return g1x_add(a, b);
}
template <int LANES>
__device__ __forceinline__ g1x tail_double(const g1x& a, uint32_t q) {
    if (LANES == 4) return g1x_double_q4(a, q);
    return g1x_double(a);
}
// tree over the block's 256 / LANES logical threads: sh[0] = sum
template <int LANES>
__device__ __forceinline__ void block_tree_sum_t(g1x* sh, uint32_t lt, uint32_t q, const g1x& mine) {
    if (q == 0) sh[lt] = mine;
    __syncthreads();
    for (uint32_t d = 128 / LANES; d >= 1; d >>= 1) {
        if (lt < d) {
            g1x r = tail_add<LANES>(sh[lt], sh[lt + d], q);
            if (q == 0) sh[lt] = r;
        }
        __syncthreads();
    }
}
// sum_{b} (b+1) * S_b over CH consecutive buckets per logical thread (running-sum trick + base * run), then a
// tree over the block: one partial sum per block.
// tot_all != null (two-level tail): chunk t's total T_t = sum of its buckets is STORED (tot_all[col][t], raw XYZZ) instead of being
// multiplied by the chunk's base; k_chunk_totals sums t * T_t over the chunks with the same running-sum trick.
template <int LANES>
__global__ void __launch_bounds__(256) k_bucket_chunks(const uint32_t* part_all, size_t part_stride, const uint32_t* cnt_all,
                                                       const uint32_t* off_all, uint32_t B, uint32_t CH, uint32_t* out_all,
                                                       uint32_t nchunks, uint32_t* tot_all) {
    constexpr uint32_t PER_BLOCK = 256 / LANES;
    __shared__ g1x sh[PER_BLOCK];
    const uint32_t col = blockIdx.y, q = threadIdx.x % LANES, lt = threadIdx.x / LANES;
    const uint32_t t = blockIdx.x * PER_BLOCK + lt;
    g1x acc = g1x_identity();
    if (t < nchunks) {
        g1x total = g1x_identity();
        const uint32_t* cnt = cnt_all + (size_t)col * B;
        const uint32_t* off = off_all + (size_t)col * (B + 4);
        const uint32_t* part = part_all + (size_t)col * part_stride * PART_WORDS;
        uint32_t base = t * CH;
        // a chunk without partial sums contributes nothing (columns of small values leave most windows empty), and inside a chunk
        // the running sum is the identity until the first non-empty bucket from the top
        if (off[min(base + CH, B)] != off[base]) {
            g1x run = g1x_identity();
            bool have = false;
            for (int j = (int)CH - 1; j >= 0; --j) {
                uint32_t b = base + (uint32_t)j;
                if (b < B) {   // the bucket's (few) partial sums are folded here: no separate reduction round for them
                    const uint32_t c = cnt[b], o = off[b];
                    for (uint32_t i = 0; i < c; ++i) run = tail_add<LANES>(run, g1x_load_loose(part + (size_t)(o + i) * PART_WORDS), q);
                    have |= c != 0;
                }
                if (have) acc = tail_add<LANES>(acc, run, q);
            }
            if (tot_all) {
                total = run;
            } else {   // + base * run
                g1x d = run;
                uint32_t m = base;
                while (m) {
                    if (m & 1) acc = tail_add<LANES>(acc, d, q);
                    m >>= 1;
                    if (m) d = tail_double<LANES>(d, q);
                }
            }
        }
        if (tot_all && q == 0) g1x_store_raw(tot_all + ((size_t)col * nchunks + t) * 32, total);
    }
    block_tree_sum_t<LANES>(sh, lt, q, acc);
    if (threadIdx.x == 0) g1x_store_raw(out_all + ((size_t)col * gridDim.x + blockIdx.x) * 32, sh[0]);
}
// Second level of the tail: sum_t t * T_t over the chunk totals, CH2 consecutive chunks per logical thread — the running sums give
// sum_j j * T_(base + j), the super-chunk's base is applied by double-and-add to ITS total: nchunks / CH2 of those products instead of one
// per chunk (a quarter of the one-level tail's additions at CH = 32, half of them at CH = 4-8).  One partial sum per block.
template <int LANES>
__global__ void __launch_bounds__(256) k_chunk_totals(const uint32_t* tot_all, uint32_t nchunks, uint32_t CH2, uint32_t* out_all, uint32_t nsuper) {
    constexpr uint32_t PER_BLOCK = 256 / LANES;
    __shared__ g1x sh[PER_BLOCK];
    const uint32_t col = blockIdx.y, q = threadIdx.x % LANES, lt = threadIdx.x / LANES;
    const uint32_t s = blockIdx.x * PER_BLOCK + lt;
    g1x acc = g1x_identity();
    if (s < nsuper) {
        const uint32_t* tot = tot_all + (size_t)col * nchunks * 32;
        const uint32_t base = s * CH2;
        g1x run = g1x_identity();
        for (int j = (int)CH2 - 1; j >= 0; --j) {
            const uint32_t t = base + (uint32_t)j;
            if (t < nchunks) run = tail_add<LANES>(run, g1x_load_raw(tot + (size_t)t * 32), q);
            if (j >= 1) acc = tail_add<LANES>(acc, run, q);
        }
        g1x d = run;
        uint32_t m = base;
        while (m) {
            if (m & 1) acc = tail_add<LANES>(acc, d, q);
            m >>= 1;
            if (m) d = tail_double<LANES>(d, q);
        }
    }
    block_tree_sum_t<LANES>(sh, lt, q, acc);
    if (threadIdx.x == 0) g1x_store_raw(out_all + ((size_t)col * gridDim.x + blockIdx.x) * 32, sh[0]);
}

// in2_all != null (two-level tail): out = sum(in) + 2^shift * sum(in2) — the chunk totals' weighted sum, times the chunk length
__global__ void __launch_bounds__(256) k_final_sum(const uint32_t* in_all, uint32_t count, uint32_t* out_all, const uint32_t* in2_all, uint32_t count2,
                                                   uint32_t shift) {
    __shared__ g1x sh[64];
    const uint32_t col = blockIdx.x, q = threadIdx.x & 3, lt = threadIdx.x >> 2;
    g1x second = g1x_identity();
    if (in2_all) {
        const uint32_t* in2 = in2_all + (size_t)col * count2 * 32;
        g1x acc2 = g1x_identity();
        for (uint32_t i = lt; i < count2; i += 64) acc2 = g1x_add_q4(acc2, g1x_load_raw(in2 + (size_t)i * 32), q);
        block_tree_sum_t<4>(sh, lt, q, acc2);
        if (lt == 0) {
            second = sh[0];
            for (uint32_t j = 0; j < shift; ++j) second = g1x_double_q4(second, q);
        }
        __syncthreads();   // sh[0] is read above and rewritten by the tree below
    }
    const uint32_t* in = in_all + (size_t)col * count * 32;
    g1x acc = lt == 0 ? second : g1x_identity();
    for (uint32_t i = lt; i < count; i += 64) acc = g1x_add_q4(acc, g1x_load_raw(in + (size_t)i * 32), q);
    block_tree_sum_t<4>(sh, lt, q, acc);
    if (threadIdx.x == 0) g1j_store_abi(out_all + (size_t)col * 24, g1x_to_jacobian(sh[0]));   // the ABI result: halo2curves G1 (R = 2^256)
}
// self-check of the quad-cooperative operations against the one-lane ones (tests/test_gpu_msm.py): for pair i of XYZZ points
// (a, b): out[i] = {a + b, 2a} by g1x_add / g1x_double and by the quad versions, 4 x 128 B
__global__ void k_quad_selfcheck(const uint32_t* pts, uint32_t npairs, uint32_t* out) {
    uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t i = lane >> 2, q = lane & 3;
    if (i >= npairs) return;
    g1x a = g1x_load_raw(pts + (size_t)i * 64), b = g1x_load_raw(pts + (size_t)i * 64 + 32);
    g1x s4 = g1x_add_q4(a, b, q), d4 = g1x_double_q4(a, q);
    if (q == 0) {
        g1x_store_raw(out + (size_t)i * 128, g1x_add(a, b));
        g1x_store_raw(out + (size_t)i * 128 + 32, g1x_double(a));
    }
    if (q == (i & 3)) {   // any lane of the quad holds the result
        g1x_store_raw(out + (size_t)i * 128 + 64, s4);
        g1x_store_raw(out + (size_t)i * 128 + 96, d4);
    }
}
extern "C" int zkt_quad_selfcheck(zkhip_ctx* ctx, const void* d_pts, uint32_t npairs, void* d_out) {
    hipLaunchKernelGGL(k_quad_selfcheck, dim3(div_up((size_t)npairs * 4, 256)), dim3(256), 0, ctx->stream, (const uint32_t*)d_pts, npairs, (uint32_t*)d_out);
    ZK_LAUNCH_CHECK();
    return ZKHIP_OK;
}

__global__ void k_set_identity(uint32_t* out_all, uint32_t ncols) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < ncols) g1j_store_abi(out_all + (size_t)t * 24, g1j_identity());
}

// ------------------------------------------------------------------ host driver
static uint32_t ilog2_u32(uint32_t v) { uint32_t l = 0; while ((1u << (l + 1)) <= v) ++l; return l; }
// columns hold at least first + n scalars; scalar first + i pairs with base first + i (indices LOCAL to the handles' tables).
// What the accumulation (round 0 + the reduction rounds for heavy buckets) leaves behind: per column and bucket b, cnt[b] partial sums (loose XYZZ, PART_WORDS
// words each) at p[off[b] ..]; the tail folds them, weights bucket b by b + 1 and sums (msm_tail).  empty: every digit of every column was zero.
struct MsmPartials { const uint32_t* p; const uint32_t* cnt; const uint32_t* off; size_t stride; uint32_t B; bool empty; };
static int msm_tail(zkhip_ctx* ctx, size_t ncols, const MsmPartials& P, void* d_out);

// digits, sort by bucket, plan, bucket accumulation and the reduction rounds of ncols columns over points [first, first + n) -> the buckets' partial sums
static int msm_partials(zkhip_ctx* ctx, const zkhip_srs* const* srs_per_col, const void* const* d_cols_host, size_t ncols, size_t first, size_t n,
                        MsmPartials* P) {
    const zkhip_srs* srs = srs_per_col[0];
    for (size_t j = 0; j < ncols; ++j) {
        const zkhip_srs* q = srs_per_col[j];
        if (!q) { set_error("zkhip_msm: the SRS of column %zu is null", j); return ZKHIP_EINVAL; }
        if (q->n != srs->n || q->c != srs->c) { set_error("zkhip_msm: the SRS of column %zu has a different size", j); return ZKHIP_EINVAL; }
    }
    if (first + n > srs->n) { set_error("zkhip_msm: range [%zu, %zu) exceeds the %zu bases loaded", first, first + n, srs->n); return ZKHIP_EINVAL; }
    hipStream_t st = ctx->stream;
    const uint32_t c = srs->c, W = srs->W, B = srs->B;
    *P = MsmPartials{nullptr, nullptr, nullptr, 0, B, true};
    if (n == 0) return ZKHIP_OK;
    const size_t items = n * W;
    const uint32_t seg0_min = 8;
    // Round-0 segment length depends on the problem size only: aim for ~2 waves per SIMD over the chip.
    uint32_t seg = seg0_min;
    {
        // about four waves per SIMD over the launch, as a power of two (L = 24 / 40 cost 2-3 % of a 2^17 proof against 16 / 32: the
        // lane index arithmetic divides by L) and at least 16 (at 8 a single column's buckets collect more than 8 partial sums each
        // and every MSM pays reduction rounds)
        size_t target_threads = (size_t)256 * 4 * 64 * 4;
        size_t sgl = (ncols * items) / target_threads;
        if (sgl >= 8) {   // smaller launches are latency chains: keep the lanes short (8)
            seg = 16;
            const uint32_t cap = n >= ((size_t)1 << 20) ? 64 : 32;   // measured: 2^18-2^19 prefer 16-32 (-3 %), 2^20-2^22 64
            while (seg < cap && (size_t)seg * 2 <= sgl) seg *= 2;
        }
        { int v = ctx->opt.msm_seg; if (v >= (int)seg0_min && v <= 256) seg = (uint32_t)v; }
    }
    // scratch
    void *d_colptrs, *d_zero, *d_off, *d_tmp_entry, *d_tmp_key, *d_entries, *d_max, *d_cntA, *d_cntB, *d_offA, *d_offB, *d_pA, *d_pB;
    // Partial sums per column: one per (round-0 lane, bucket it touches).  A lane covers l consecutive sorted entries, l = lane_len(total, items,
    // seg, ncols) >= total * seg / items (and >= 2), so there are at most total / l + B <= items / seg + B of them whatever the column's density;
    // the reduction rounds only shrink that.  (Until round 4 the two buffers were sized for seg = 8: 11.3 GiB for a five-column batch at k = 22,
    // of which 2.1 GiB could ever be written.)
    const size_t pstride0 = items / seg + B + 64;
    SortGeom g;
    g.c = c; g.W = W; g.B = B;
    g.R = 1;
    while (g.R < SORT_COPIES && (size_t)g.R * 512 <= div_up(n, 256)) g.R *= 2;   // ~256+ workgroups per copy
    { int v = ctx->opt.sort_copies; if (v >= 1 && v <= (int)SORT_COPIES && (v & (v - 1)) == 0) g.R = (uint32_t)v; }
    const uint32_t KB = c - 1;
    g.HB = KB >= 18 ? 8 : (KB > 8 ? 7 : KB / 2);   // 256 partitions from c = 19 (measured: digits -14 % at 2^20, -8 % at 2^21)
    { int v = ctx->opt.sort_hb; if (v >= 1 && v <= 8 && v < (int)KB && (int)KB - v <= 11) g.HB = (uint32_t)v; }
    g.LB = KB - g.HB;
    g.P = 1u << g.HB;
    // Measured at 2^22 (2^11 bins): tiles of 16k / 32k / 64k pairs, which lengthen the scatter's contiguous runs from 8 to
    // 32-128 bytes, are 18-26 % SLOWER than 4096-pair tiles — the low pass is bound by its LDS rank atomics, not by run length.
    g.tile = g.LB >= 11 ? 2 * SORT_TILE : SORT_TILE;   // 2048 bins (c = 19, n >= 2^20): 8192-pair tiles, 16-byte runs (digits -13 % at 2^22)
    { int v = ctx->opt.sort_tile; if (v >= 1024 && v <= (1 << 20)) g.tile = (uint32_t)v; }
    if (g.LB > 11) { set_error("zkhip_msm: window c = %u unsupported by the sort (max 20)", c); return ZKHIP_EINVAL; }
    ZK_TRY(ctx->get_scratch("msm_colptrs", 2 * ncols * sizeof(void*), &d_colptrs));
    // zeroed every call: part_cnt[SORT_MAXP] + part_cursor[SORT_MAXP] + cnt[B] + cursor[B] per column
    const size_t zero_words = ncols * (2 * (size_t)SORT_MAXP * SORT_COPIES + 2 * (size_t)B);
    ZK_TRY(ctx->get_scratch("msm_zero", zero_words * 4, &d_zero));
    uint32_t* d_part_cnt = (uint32_t*)d_zero;
    uint32_t* d_part_cursor = d_part_cnt + ncols * SORT_MAXP * SORT_COPIES;
    uint32_t* d_cnt = d_part_cursor + ncols * SORT_MAXP * SORT_COPIES;
    uint32_t* d_cursor = d_cnt + ncols * (size_t)B;
    void* d_part;
    ZK_TRY(ctx->get_scratch("msm_part", ncols * 2 * SORT_PSTRIDE * 4, &d_part));   // part_off + tile_start, SORT_PSTRIDE each per column
    uint32_t* d_part_off = (uint32_t*)d_part;
    uint32_t* d_tile_start = d_part_off + ncols * SORT_PSTRIDE;
    ZK_TRY(ctx->get_scratch("msm_off", ncols * (B + 4) * 4, &d_off));
    ZK_TRY(ctx->get_scratch("msm_tmp_entry", ncols * items * 4, &d_tmp_entry));
    ZK_TRY(ctx->get_scratch("msm_tmp_key", ncols * items * 2, &d_tmp_key));
    ZK_TRY(ctx->get_scratch("msm_entries", ncols * items * 4, &d_entries));
    ZK_TRY(ctx->get_scratch("msm_max", ncols * 4, &d_max));
    ZK_TRY(ctx->get_scratch("msm_cntA", ncols * B * 4, &d_cntA));
    ZK_TRY(ctx->get_scratch("msm_cntB", ncols * B * 4, &d_cntB));
    ZK_TRY(ctx->get_scratch("msm_offA", ncols * (B + 4) * 4, &d_offA));
    ZK_TRY(ctx->get_scratch("msm_offB", ncols * (B + 4) * 4, &d_offB));
    ZK_TRY(ctx->get_scratch("msm_pA", ncols * pstride0 * PART_WORDS * 4, &d_pA));
    ZK_TRY(ctx->get_scratch("msm_pB", ncols * pstride0 * PART_WORDS * 4, &d_pB));
    std::vector<const void*> h_ptrs(2 * ncols);
    for (size_t j = 0; j < ncols; ++j) { h_ptrs[j] = d_cols_host[j]; h_ptrs[ncols + j] = srs_per_col[j]->d_table; }
    ColPtrs cp_scalars, cp_tables;
    if (ncols <= 16) {
        cp_scalars.dev = cp_tables.dev = nullptr;
        for (size_t j = 0; j < ncols; ++j) { cp_scalars.val[j] = (const uint32_t*)h_ptrs[j]; cp_tables.val[j] = (const uint32_t*)h_ptrs[ncols + j]; }
    } else {
        ZK_TRY(ctx->upload(d_colptrs, h_ptrs.data(), 2 * ncols * sizeof(void*)));
        cp_scalars.dev = (const uint32_t* const*)d_colptrs;
        cp_tables.dev = (const uint32_t* const*)((const void**)d_colptrs + ncols);
    }
    ZK_HIP(hipMemsetAsync(d_zero, 0, zero_words * 4, st));
    dim3 gn(div_up(n, 256), (unsigned)ncols);
    dim3 gt(div_up(items, g.tile) + g.P, (unsigned)ncols);
    // the "heaviest bucket" read-back: the plan kernel stores it straight into pinned host memory (slot at 1 KiB) when it fits
    std::vector<uint32_t> h_max_big;
    uint32_t* h_max = (uint32_t*)ctx->h_pinned;
    const bool max_pinned = ncols * 8 <= 1024;   // 8-byte slots: (tag of this call << 32) | value, polled by the host
    const uint32_t max_tag = max_pinned ? (++ctx->max_seq ? ctx->max_seq : ++ctx->max_seq) : 0;
    if (max_pinned) { h_max = (uint32_t*)((char*)ctx->h_pinned + 1024); d_max = h_max; }
    else if (ncols * 4 > zkhip_ctx::PINNED_BYTES) { h_max_big.resize(ncols); h_max = h_max_big.data(); }
    { ProfScope ps(ctx, "msm_digits");
    hipLaunchKernelGGL(k_sort_hi<false>, gn, dim3(256), 0, st, cp_scalars, n, first, srs->n, g, d_part_cnt, d_part_cursor,
                       (const uint32_t*)d_part_off, (uint32_t*)d_tmp_entry, (uint16_t*)d_tmp_key, items);
    hipLaunchKernelGGL(k_part_scan, dim3((unsigned)ncols), dim3(SORT_MAXP), 0, st, d_part_cnt, g.P, g.R, g.tile, d_part_off, d_tile_start);
    hipLaunchKernelGGL(k_sort_hi<true>, gn, dim3(256), g.W <= 24 ? (size_t)256 * g.W * 7 + 16 : 0, st, cp_scalars, n, first, srs->n, g, d_part_cnt, d_part_cursor,
                       (const uint32_t*)d_part_off, (uint32_t*)d_tmp_entry, (uint16_t*)d_tmp_key, items);
    hipLaunchKernelGGL(k_sort_lo_count, gt, dim3(256), 0, st, (const uint32_t*)d_part_off, (const uint32_t*)d_tile_start, g,
                       (const uint16_t*)d_tmp_key, items, d_cnt); }
    // plan: bucket offsets, then the number of round-0 lanes touching each bucket (= its partial sums) and their offsets
    const uint32_t L = seg;
    const bool adaptive_L = ctx->opt.msm_adaptive_l != 0;
    { ProfScope ps(ctx, "msm_plan");
    ZK_TRY(launch_plan(ctx, (unsigned)ncols, (const uint32_t*)d_cnt, B, 1, (uint32_t*)d_off, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr));
    // second scan, in lane mode: scans npart[b] (computed on the fly from cnt and off), writes it to cntA, its offsets to offA
    ZK_TRY(launch_plan(ctx, (unsigned)ncols, (const uint32_t*)d_cnt, B, 1, (uint32_t*)d_offA, (uint32_t*)d_cntA, (uint32_t*)nullptr, (uint32_t*)d_max,
                       (const uint32_t*)d_off, L, max_tag, adaptive_L ? (uint32_t)items : 0u)); }
    if (!max_pinned) {
        ZK_HIP(hipMemcpyAsync(h_max, d_max, ncols * 4, hipMemcpyDeviceToHost, st));
        ZK_HIP(hipEventRecord(ctx->ev_read, st));
    }
    { ProfScope ps(ctx, "msm_digits");
    {
        const size_t lds = (size_t)3 * (1u << g.LB) * 4 + (size_t)g.tile * 6;
        static std::once_flag attr_once;
        hipError_t attr_err = hipSuccess;
        std::call_once(attr_once, [&] {
            attr_err = hipFuncSetAttribute((const void*)k_sort_lo_staged, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            if (attr_err == hipSuccess) attr_err = hipFuncSetAttribute((const void*)k_sort_lo_staged16<32, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            if (attr_err == hipSuccess) attr_err = hipFuncSetAttribute((const void*)k_sort_lo_staged16<8, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            if (attr_err == hipSuccess) attr_err = hipFuncSetAttribute((const void*)k_sort_lo_staged16<16, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            if (attr_err == hipSuccess) attr_err = hipFuncSetAttribute((const void*)k_sort_lo_staged16<16, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            if (attr_err == hipSuccess) attr_err = hipFuncSetAttribute((const void*)k_sort_lo_staged16<4, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        });
        ZK_HIP(attr_err);
        const bool one_atomic = ctx->opt.sort_one_atomic != 0;
        if (g.tile == 8192 && one_atomic && (ctx->opt.sort_wide == 1 || ctx->opt.sort_wide < 0))
            hipLaunchKernelGGL((k_sort_lo_staged16<8, 1024>), gt, dim3(1024), lds, st, (const uint32_t*)d_part_off, (const uint32_t*)d_tile_start, g,
                               (const uint32_t*)d_tmp_entry, (const uint16_t*)d_tmp_key, items, (const uint32_t*)d_off, d_cursor, (uint32_t*)d_entries);
        else if (g.tile == 16384 && one_atomic && ctx->opt.sort_wide == 2)
            hipLaunchKernelGGL((k_sort_lo_staged16<16, 1024>), gt, dim3(1024), lds, st, (const uint32_t*)d_part_off, (const uint32_t*)d_tile_start, g,
                               (const uint32_t*)d_tmp_entry, (const uint16_t*)d_tmp_key, items, (const uint32_t*)d_off, d_cursor, (uint32_t*)d_entries);
        else if (g.tile == 8192 && one_atomic && ctx->opt.sort_wide == 3)
            hipLaunchKernelGGL((k_sort_lo_staged16<16, 512>), gt, dim3(512), lds, st, (const uint32_t*)d_part_off, (const uint32_t*)d_tile_start, g,
                               (const uint32_t*)d_tmp_entry, (const uint16_t*)d_tmp_key, items, (const uint32_t*)d_off, d_cursor, (uint32_t*)d_entries);
        else if (g.tile == 4096 && one_atomic && ctx->opt.sort_wide == 4)
            hipLaunchKernelGGL((k_sort_lo_staged16<4, 1024>), gt, dim3(1024), lds, st, (const uint32_t*)d_part_off, (const uint32_t*)d_tile_start, g,
                               (const uint32_t*)d_tmp_entry, (const uint16_t*)d_tmp_key, items, (const uint32_t*)d_off, d_cursor, (uint32_t*)d_entries);
        else if (g.tile == 8192 && one_atomic)
            hipLaunchKernelGGL((k_sort_lo_staged16<32, 256>), gt, dim3(256), lds, st, (const uint32_t*)d_part_off, (const uint32_t*)d_tile_start, g,
                               (const uint32_t*)d_tmp_entry, (const uint16_t*)d_tmp_key, items, (const uint32_t*)d_off, d_cursor, (uint32_t*)d_entries);
        else if (g.tile == SORT_TILE && one_atomic)
            hipLaunchKernelGGL((k_sort_lo_staged16<16, 256>), gt, dim3(256), lds, st, (const uint32_t*)d_part_off, (const uint32_t*)d_tile_start, g,
                               (const uint32_t*)d_tmp_entry, (const uint16_t*)d_tmp_key, items, (const uint32_t*)d_off, d_cursor, (uint32_t*)d_entries);
        else
        hipLaunchKernelGGL(k_sort_lo_staged, gt, dim3(256), lds, st, (const uint32_t*)d_part_off, (const uint32_t*)d_tile_start, g,
                           (const uint32_t*)d_tmp_entry, (const uint16_t*)d_tmp_key, items, (const uint32_t*)d_off, d_cursor, (uint32_t*)d_entries);
    } }
    // round 0 does not need the maximum: it is issued before the host waits for it
    { ProfScope ps(ctx, "msm_accum_affine");
    hipLaunchKernelGGL(k_accum_affine, dim3(div_up(div_up(items, L), 256), (unsigned)ncols), dim3(256), 0, st,
                       cp_tables, (const uint32_t*)d_entries, items, (const uint32_t*)d_off,
                       (const uint32_t*)d_offA, B, L, (uint32_t*)d_pA, pstride0, adaptive_L ? 1u : 0u); }
    if (ctx->accum_mark) {   // a caller wants to start overlapped work when the throughput-bound part of this MSM is over
        ZK_HIP(hipEventRecord(ctx->accum_mark, st));
        ctx->accum_mark = nullptr;
    }
    ZK_LAUNCH_CHECK();
    uint32_t maxcnt = 0;   // most partial sums in one bucket
    if (max_pinned) {
        // poll the tagged slots: no event, no marker packet between the plan and the scatter (a marker costs ~15 us of idle GPU)
        const volatile uint64_t* slots = (const volatile uint64_t*)h_max;
        for (size_t j = 0; j < ncols; ++j) {
            uint64_t v = slots[j];
            for (uint64_t spin = 0; (uint32_t)(v >> 32) != max_tag; ++spin) {
                if ((spin & 0xFFFFF) == 0xFFFFF) {   // a failed launch would never publish: look at the stream now and then
                    hipError_t e = hipStreamQuery(st);
                    if (e != hipSuccess && e != hipErrorNotReady) ZK_HIP(e);
                    if (e == hipSuccess && (uint32_t)(slots[j] >> 32) != max_tag) { set_error("msm: plan read-back never arrived"); return ZKHIP_EHIP; }
                }
                v = slots[j];
            }
            maxcnt = std::max(maxcnt, (uint32_t)v);
        }
    } else {
        if (h_max_big.empty()) ZK_HIP(event_wait(ctx, ctx->ev_read));   // only the read-back: the scatter and round 0 are still running
        else ZK_HIP(stream_wait(ctx, st));                               // pageable destination: wait for everything
        for (size_t j = 0; j < ncols; ++j) maxcnt = std::max(maxcnt, h_max[j]);
    }
    if (maxcnt == 0) return ZKHIP_OK;      // (P->empty)
    if (ctx->opt.msm_debug) fprintf(stderr, "msm: n=%zu ncols=%zu c=%u W=%u L=%u max partials per bucket=%u\n", n, ncols, c, W, L, maxcnt);
    const uint32_t* cur_cnt = (const uint32_t*)d_cntA;
    const uint32_t* cur_off = (const uint32_t*)d_offA;
    uint32_t* nxt_cnt = (uint32_t*)d_cntB;
    uint32_t* nxt_off = (uint32_t*)d_offB;
    uint32_t* cur_p = (uint32_t*)d_pA;
    uint32_t* nxt_p = (uint32_t*)d_pB;
    size_t bound = items / L + B + 1;
    if (bound > pstride0) bound = pstride0;
    // the tail folds up to tail_parts partial sums per bucket itself; heavier buckets (skewed scalars) go through reduction rounds
    // From 2^20 points a reduction round pays whenever a bucket holds more than 4 partial sums (k_accum_jac folds them with every
    // lane busy; the tail's quads would do it inside their dependent chains: 11.5 against 3.0 + 5.3 ms per k = 22 proof); below, where
    // the round is a latency chain of its own, only skewed columns get one.
    uint32_t tail_parts = n >= ((size_t)1 << 20) ? 4 : 8;
    { int v = ctx->opt.msm_tailparts; if (v >= 1 && v <= 64) tail_parts = (uint32_t)v; }
    while (maxcnt > tail_parts) {
        seg = maxcnt <= 16 ? maxcnt : 8;
        { ProfScope ps(ctx, "msm_plan");
        ZK_TRY(launch_plan(ctx, (unsigned)ncols, cur_cnt, B, seg, (uint32_t*)nullptr, nxt_cnt, nxt_off, (uint32_t*)nullptr)); }
        size_t nb = bound / seg + B + 1;
        if (nb > pstride0) nb = pstride0;
        { ProfScope ps(ctx, "msm_accum_jac");
        if (nb * ncols < (size_t)160 * 1024)   // under ~2.5 waves per SIMD the round is a latency chain: 4 lanes per segment
            hipLaunchKernelGGL(k_accum_jac_q4, dim3(div_up(nb * 4, 256), (unsigned)ncols), dim3(256), 0, st, (const uint32_t*)cur_p, pstride0,
                               cur_cnt, cur_off, (const uint32_t*)nxt_off, B, seg, nxt_p, pstride0);
        else
            hipLaunchKernelGGL(k_accum_jac, dim3(div_up(nb, 256), (unsigned)ncols), dim3(256), 0, st, (const uint32_t*)cur_p, pstride0,
                               cur_cnt, cur_off, (const uint32_t*)nxt_off, B, seg, nxt_p, pstride0); }
        bound = nb;
        std::swap(cur_p, nxt_p);
        const uint32_t* tc = cur_cnt; const uint32_t* to = cur_off;
        cur_cnt = nxt_cnt; cur_off = nxt_off;
        nxt_cnt = (uint32_t*)tc; nxt_off = (uint32_t*)to;
        maxcnt = (maxcnt + seg - 1) / seg;
    }
    if (ctx->prof_on && ctx->prof_only.empty()) {
        // profiling passes only: the number of (non-zero digit, point) pairs the accumulation really processed (zero digits are
        // skipped, so bit / small-valued witness columns have far fewer than n W) — off[B] of every column, one small read-back
        std::vector<uint32_t> tot(ncols);
        for (size_t j = 0; j < ncols; ++j)
            ZK_HIP(hipMemcpyAsync(&tot[j], (const uint32_t*)d_off + j * (B + 4) + B, 4, hipMemcpyDeviceToHost, st));
        ZK_HIP(hipStreamSynchronize(st));
        for (size_t j = 0; j < ncols; ++j) ctx->prof_msm_pairs += tot[j];
        ctx->prof_msm_dense_pairs += (uint64_t)ncols * items;
    }
    *P = MsmPartials{cur_p, cur_cnt, cur_off, pstride0, B, false};
    return ZKHIP_OK;
}

// The bucket reduction ("tail"): out[col] = sum_b (b + 1) * (sum of bucket b's partial sums), by chunks of CH buckets with running sums.
static int msm_tail(zkhip_ctx* ctx, size_t ncols, const MsmPartials& P, void* d_out) {
    hipStream_t st = ctx->stream;
    const uint32_t B = P.B;
    const uint32_t* cur_p = P.p;
    const uint32_t* cur_cnt = P.cnt;
    const uint32_t* cur_off = P.off;
    const size_t pstride0 = P.stride;
    void* d_chunks;
    uint32_t CH = B > 8192 ? std::min<uint32_t>(B / 8192, 32) : 1;   // 2^19 buckets (c = 20): 32 beats 64 by 0.4 ms per k = 22 proof, 16 equals 32
    { int v = ctx->opt.msm_ch; if (v >= 1 && v <= 256) CH = (uint32_t)v; }
    uint32_t nchunks = (B + CH - 1) / CH;
    // a wide batch has enough chunks to fill the chip with one lane each; otherwise four lanes share every point operation
    bool wide_tail = (size_t)nchunks * ncols >= (size_t)48 * 1024;   // measured crossover: 6-8 columns at 8192 chunks
    if (ctx->opt.msm_widetail >= 0) wide_tail = ctx->opt.msm_widetail != 0;
    uint32_t nchunk_blocks = div_up(nchunks, wide_tail ? 256 : 64);
    ZK_TRY(ctx->get_scratch("msm_chunks", ncols * (size_t)nchunk_blocks * 128, &d_chunks));
    // Two-level tail (round 4): the chunks' bases are applied to the chunk TOTALS by a second running-sum pass instead of one double-and-add
    // per chunk (~21 of a chunk's 2 CH + 21 point operations).  It removes work, not depth — it adds a launch — so it is for the wide
    // batches, whose tail runs beside the other stream's transforms; the one- and two-column MSMs of the multi-open run alone and keep
    // the shorter one-level chain.
    const uint32_t CH2 = 8;
    bool two_level = wide_tail && nchunks >= 4096 && (CH & (CH - 1)) == 0;
    if (ctx->opt.msm_tail2 >= 0) two_level = ctx->opt.msm_tail2 != 0 && nchunks >= 2 * CH2 && (CH & (CH - 1)) == 0;
    const uint32_t nsuper = div_up(nchunks, CH2);
    const bool wide2 = (size_t)nsuper * ncols >= (size_t)48 * 1024;
    const uint32_t nblk2 = div_up(nsuper, wide2 ? 256 : 64);
    void *d_tot = nullptr, *d_chunks2 = nullptr;
    if (two_level) {
        ZK_TRY(ctx->get_scratch("msm_tot", ncols * (size_t)nchunks * 128, &d_tot));
        ZK_TRY(ctx->get_scratch("msm_chunks2", ncols * (size_t)nblk2 * 128, &d_chunks2));
    }

    { ProfScope ps(ctx, "msm_tail");
    if (wide_tail)
        hipLaunchKernelGGL(k_bucket_chunks<1>, dim3(nchunk_blocks, (unsigned)ncols), dim3(256), 0, st, (const uint32_t*)cur_p, pstride0,
                           cur_cnt, cur_off, B, CH, (uint32_t*)d_chunks, nchunks, (uint32_t*)d_tot);
    else
        hipLaunchKernelGGL(k_bucket_chunks<4>, dim3(nchunk_blocks, (unsigned)ncols), dim3(256), 0, st, (const uint32_t*)cur_p, pstride0,
                           cur_cnt, cur_off, B, CH, (uint32_t*)d_chunks, nchunks, (uint32_t*)d_tot);
    if (two_level) {
        if (wide2)
            hipLaunchKernelGGL(k_chunk_totals<1>, dim3(nblk2, (unsigned)ncols), dim3(256), 0, st, (const uint32_t*)d_tot, nchunks, CH2, (uint32_t*)d_chunks2, nsuper);
        else
            hipLaunchKernelGGL(k_chunk_totals<4>, dim3(nblk2, (unsigned)ncols), dim3(256), 0, st, (const uint32_t*)d_tot, nchunks, CH2, (uint32_t*)d_chunks2, nsuper);
    }
    hipLaunchKernelGGL(k_final_sum, dim3((unsigned)ncols), dim3(256), 0, st, (const uint32_t*)d_chunks, nchunk_blocks, (uint32_t*)d_out,
                       (const uint32_t*)d_chunks2, nblk2, ilog2_u32(CH)); }
    ZK_LAUNCH_CHECK();
    return ZKHIP_OK;
}

static int msm_local(zkhip_ctx* ctx, const zkhip_srs* const* srs_per_col, const void* const* d_cols_host, size_t ncols, size_t first, size_t n,
                     void* d_out) {
    if (ncols == 0) return ZKHIP_OK;
    MsmPartials P;
    ZK_TRY(msm_partials(ctx, srs_per_col, d_cols_host, ncols, first, n, &P));
    if (P.empty) {
        hipLaunchKernelGGL(k_set_identity, dim3(div_up(ncols, 64)), dim3(64), 0, ctx->stream, (uint32_t*)d_out, (uint32_t)ncols);
        ZK_LAUNCH_CHECK();
        return ZKHIP_OK;
    }
    return msm_tail(ctx, ncols, P, d_out);
}

// zkhip_msm_g1's pipelined form: the K chunks' partial sums of ONE column, bucket by bucket, folded into one dense set (bucket b's single sum at slot b)
// so that the chunks share ONE tail.  One thread per bucket; B + 1 threads (the last writes the closing offset).
struct MergeSrc { const uint32_t* p; const uint32_t* cnt; const uint32_t* off; };
struct MergeArgs { MergeSrc s[16]; uint32_t K; };
__global__ void __launch_bounds__(256) k_merge_buckets(const MergeArgs A, uint32_t B, uint32_t* out_p, uint32_t* out_cnt, uint32_t* out_off) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b > B) return;
    out_off[b] = b;
    if (b == B) return;
    g1x acc = g1x_identity();
    bool have = false;
    for (uint32_t j = 0; j < A.K; ++j) {
        if (!A.s[j].p) continue;      // an all-zero chunk
        const uint32_t c = A.s[j].cnt[b], o = A.s[j].off[b];
        for (uint32_t i = 0; i < c; ++i) {
            const g1x v = g1x_load_loose(A.s[j].p + (size_t)(o + i) * PART_WORDS);
            acc = have ? g1x_add(acc, v) : v;
            have = true;
        }
    }
    out_cnt[b] = have ? 1u : 0u;
    if (have) g1x_store_loose(out_p + (size_t)b * PART_WORDS, acc);
}


// GLOBAL point indices.  Over whole-SRS handles this is msm_local.  Over point-range shards (zkhip_kzg_setup_range /
// zkhip_srs_load_range) every rank sums the part of [first, first + n) its tables hold — scalar pointers shifted so that local base 0
// meets its scalar — and, when the context has a communicator, the partial sums are all-gathered and folded (collective call).
static int msm_run(zkhip_ctx* ctx, const zkhip_srs* const* srs_per_col, const void* const* d_cols_host, size_t ncols, size_t first, size_t n,
                   void* d_out) {
    if (!ctx || !srs_per_col || !d_cols_host || !d_out) { set_error("zkhip_msm: null argument"); return ZKHIP_EINVAL; }
    if (ncols == 0) return ZKHIP_OK;
    const zkhip_srs* srs = srs_per_col[0];
    if (!srs) { set_error("zkhip_msm: null SRS for column 0"); return ZKHIP_EINVAL; }
    bool sharded = false;
    for (size_t j = 0; j < ncols; ++j) {
        const zkhip_srs* q = srs_per_col[j];
        if (!q) { set_error("zkhip_msm: the SRS of column %zu is null", j); return ZKHIP_EINVAL; }
        if (q->first0 != srs->first0 || q->n != srs->n || q->n_total != srs->n_total) { set_error("zkhip_msm: the SRS of column %zu covers a different range", j); return ZKHIP_EINVAL; }
        sharded |= q->n_total != q->n;
    }
    if (first + n > srs->n_total) { set_error("zkhip_msm: range [%zu, %zu) exceeds the %zu bases of the SRS", first, first + n, srs->n_total); return ZKHIP_EINVAL; }
    if (!sharded && ctx->comm.nranks > 1 && ctx->comm.shard_columns) {
        // whole-SRS handles on a context with a communicator: the batch is split by COLUMN (independent commitments
        // round-robin over the GPUs — the better split while one MSM cannot fill several GPUs, k <= 19).  Rank r computes columns
        // r, r + N, ... completely; the other columns of its partial vector are the identity, so the same all-gather + fold applies.
        const size_t NR = (size_t)ctx->comm.nranks, RK = (size_t)ctx->comm.rank;
        void *d_part, *d_mine;
        ZK_TRY(ctx->get_scratch("msm_shard_part", ncols * 96, &d_part));
        ZK_TRY(ctx->get_scratch("msm_shard_mine", ncols * 96, &d_mine));
        std::vector<const zkhip_srs*> my_srs;
        std::vector<const void*> my_cols;
        for (size_t j = RK; j < ncols; j += NR) { my_srs.push_back(srs_per_col[j]); my_cols.push_back(d_cols_host[j]); }
        hipLaunchKernelGGL(k_set_identity, dim3(div_up(ncols, 64)), dim3(64), 0, ctx->stream, (uint32_t*)d_part, (uint32_t)ncols);
        ZK_LAUNCH_CHECK();
        if (!my_cols.empty()) {
            ZK_TRY(msm_local(ctx, my_srs.data(), my_cols.data(), my_cols.size(), first, n, d_mine));
            for (size_t i = 0, j = RK; j < ncols; ++i, j += NR)
                ZK_HIP(hipMemcpyAsync((char*)d_part + j * 96, (const char*)d_mine + i * 96, 96, hipMemcpyDeviceToDevice, ctx->stream));
        }
        return zk::comm_fold_partials(ctx, d_part, ncols, d_out);
    }
    if (!sharded) return msm_local(ctx, srs_per_col, d_cols_host, ncols, first, n, d_out);
    const size_t lo = std::max(first, srs->first0), hi = std::min(first + n, srs->first0 + srs->n);
    const size_t cnt = hi > lo ? hi - lo : 0;
    const bool collective = ctx->comm.nranks > 1;
    if (!collective && cnt != n) { set_error("zkhip_msm: range [%zu, %zu) is not inside this shard [%zu, %zu) and the context has no communicator", first, first + n, srs->first0, srs->first0 + srs->n); return ZKHIP_EINVAL; }
    std::vector<const void*> shifted(ncols);
    for (size_t j = 0; j < ncols; ++j) shifted[j] = (const char*)d_cols_host[j] + srs->first0 * 32;
    void* d_part = d_out;
    if (collective) ZK_TRY(ctx->get_scratch("msm_shard_part", ncols * 96, &d_part));
    ZK_TRY(msm_local(ctx, srs_per_col, shifted.data(), ncols, cnt ? lo - srs->first0 : 0, cnt, d_part));
    if (collective) ZK_TRY(zk::comm_fold_partials(ctx, d_part, ncols, d_out));
    return ZKHIP_OK;
}

namespace zk {
int srs_set_range(zkhip_srs* s, size_t first, size_t n_total) { s->first0 = first; s->n_total = n_total; return ZKHIP_OK; }
}

extern "C" {

int zkhip_srs_load_range(zkhip_ctx* ctx, const uint64_t* bases_xy, size_t n_total, size_t first, size_t count, zkhip_srs** out) {
    if (!ctx || !bases_xy || !out || count == 0 || first + count > n_total) { set_error("zkhip_srs_load_range: bad argument"); return ZKHIP_EINVAL; }
    ZK_TRY(zkhip_srs_load(ctx, bases_xy, count, out));
    (*out)->first0 = first;
    (*out)->n_total = n_total;
    return ZKHIP_OK;
}
void zkhip_srs_range(const zkhip_srs* s, size_t* first, size_t* count, size_t* n_total) {
    if (first) *first = s ? s->first0 : 0;
    if (count) *count = s ? s->n : 0;
    if (n_total) *n_total = s ? s->n_total : 0;
}

int zkhip_srs_load_device(zkhip_ctx* ctx, const void* d_bases, size_t n, zkhip_srs** out) {
    if (!ctx || !d_bases || !out) { set_error("zkhip_srs_load_device: null argument"); return ZKHIP_EINVAL; }
    return srs_build(ctx, d_bases, n, out, 0);
}
int zkhip_srs_load(zkhip_ctx* ctx, const uint64_t* bases_xy, size_t n, zkhip_srs** out) {
    if (!ctx || !bases_xy || !out) { set_error("zkhip_srs_load: null argument"); return ZKHIP_EINVAL; }
    if (n == 0) { set_error("zkhip_srs_load: n = 0"); return ZKHIP_EINVAL; }
    void* d;
    ZK_TRY(ctx->get_scratch("srs_upload", n * 64, &d));
    ZK_HIP(hipMemcpyAsync(d, bases_xy, n * 64, hipMemcpyHostToDevice, ctx->stream));
    return srs_build(ctx, d, n, out, 0);
}
void zkhip_srs_free(zkhip_ctx* ctx, zkhip_srs* s) {
    if (!s) return;
    if (ctx) (void)hipStreamSynchronize(ctx->stream);
    if (s->d_table) (void)hipFree(s->d_table);
    delete s;
}
size_t zkhip_srs_len(const zkhip_srs* s) { return s ? s->n_total : 0; }
void zkhip_srs_window(const zkhip_srs* s, uint32_t* c, uint32_t* windows) {
    if (c) *c = s ? s->c : 0;
    if (windows) *windows = s ? s->W : 0;
}
int zkhip_srs_read(zkhip_ctx* ctx, const zkhip_srs* s, size_t first, size_t count, uint64_t* out_xy) {
    if (!ctx || !s || !out_xy || first < s->first0 || first + count > s->first0 + s->n) { set_error("zkhip_srs_read: range outside this handle's bases"); return ZKHIP_EINVAL; }
    ZK_HIP(hipMemcpyAsync(out_xy, (const char*)s->d_table + (first - s->first0) * 64, count * 64, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < count; ++i) g1a_store_abi(out_xy + 8 * i, g1a_load_raw(out_xy + 8 * i));  // table form -> ABI form
    return ZKHIP_OK;
}

int zkhip_msm_g1_batch_device(zkhip_ctx* ctx, const zkhip_srs* srs, const void* const* d_scalar_cols, size_t ncols, size_t n,
                              void* d_out_xyz) {
    std::vector<const zkhip_srs*> v(ncols, srs);
    return msm_run(ctx, v.data(), d_scalar_cols, ncols, 0, n, d_out_xyz);
}
int zkhip_msm_g1_multi_device(zkhip_ctx* ctx, const zkhip_srs* const* srs_per_col, const void* const* d_scalar_cols, size_t ncols,
                              size_t first, size_t count, void* d_out_xyz) {
    return msm_run(ctx, srs_per_col, d_scalar_cols, ncols, first, count, d_out_xyz);
}
int zkhip_msm_g1_batch_range_device(zkhip_ctx* ctx, const zkhip_srs* srs, const void* const* d_scalar_cols, size_t ncols, size_t first,
                                    size_t count, void* d_out_xyz) {
    std::vector<const zkhip_srs*> v(ncols, srs);
    return msm_run(ctx, v.data(), d_scalar_cols, ncols, first, count, d_out_xyz);
}

// How many pieces the upload + MSM pipeline of a host slice of n scalars is cut into (1: one upload, one MSM)
static size_t host_msm_chunks(const zkhip_ctx* ctx, const zkhip_srs* srs, size_t n) {
    if (srs->n_total != srs->n || ctx->comm.nranks > 1 || n > srs->n) return 1;      // (a point-range shard / a communicator: the collective form, one piece)
    size_t K = n >= ((size_t)1 << 21) ? 4 : n >= ((size_t)1 << 20) ? 2 : 1;
    const int v = ctx->opt.msm_host_chunks;
    if (v >= 1 && v <= 16) K = (size_t)v;
    while (K > 1 && n / K < 65536) K /= 2;
    return K;
}
// the streams and events of the pipelined host forms (created on first use)
static int host_msm_streams(zkhip_ctx* ctx, size_t nevents) {
    if (!ctx->copy_stream) {
        ZK_HIP(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        for (auto& e : ctx->copy_event) ZK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    if (!ctx->side_stream) {
        ZK_HIP(hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
        ZK_HIP(hipEventCreateWithFlags(&ctx->side_event, hipEventDisableTiming));
    }
    if (ctx->host_chunk_event.size() < nevents) {
        const size_t have = ctx->host_chunk_event.size();
        ctx->host_chunk_event.resize(nevents, nullptr);
        for (size_t j = have; j < nevents; ++j) ZK_HIP(hipEventCreateWithFlags(&ctx->host_chunk_event[j], hipEventDisableTiming));
    }
    return ZKHIP_OK;
}
// the copy stream starts behind everything already issued on the main and the side stream (the staging columns' last readers)
static int host_msm_fence(zkhip_ctx* ctx, hipStream_t main, hipEvent_t fence) {
    ZK_HIP(hipEventRecord(fence, main));
    ZK_HIP(hipStreamWaitEvent(ctx->copy_stream, fence, 0));
    ZK_HIP(hipEventRecord(ctx->side_event, ctx->side_stream));
    ZK_HIP(hipStreamWaitEvent(ctx->copy_stream, ctx->side_event, 0));
    return ZKHIP_OK;
}
// Restores what the pipelined host forms change on the context, whatever happens: the main stream, no scratch tag; on an error exit the caller's slices are no
// longer being read (the copy stream depends on nothing but the fence) — a Vec<Fr> the caller drops must not race the DMA; registered slices are unregistered.
struct HostMsmRestore {
    zkhip_ctx* c; hipStream_t s; std::vector<const void*> registered; bool ok = false;
    ~HostMsmRestore() {
        c->stream = s;
        c->scratch_tag.clear();
        if (!ok && !c->dead && c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
        for (const void* h : registered) if (hipHostUnregister((void*)h) != hipSuccess) (void)hipGetLastError();      // (never leave a sticky error behind: the next launch check would report it)
    }
    // pins the caller's pages in place for the duration of the call: an unregistered (pageable) source makes every chunked hipMemcpyAsync block the HOST for its own
    // duration (a slice that already is pinned — hipHostMalloc, a torch pinned tensor — says so: fine, its copies are asynchronous anyway)
    void pin(const void* h, size_t bytes) {
        if (std::find(registered.begin(), registered.end(), h) != registered.end()) return;      // the same slice twice in one batch
        if (hipHostRegister((void*)h, bytes, hipHostRegisterDefault) == hipSuccess) registered.push_back(h); else (void)hipGetLastError();
    }
};
// ONE host column through the K-chunk pipeline: uploads on the copy stream (issued here), chunk j's digits, sort and bucket accumulation over points [off_j, off_j + len_j)
// on the main and the side stream alternately as soon as ITS bytes have landed, each chunk in its own scratch (scratch_tag), the chunks' bucket sums folded bucket by bucket
// (k_merge_buckets) and ONE tail.  Everything is enqueued; the result lands in d_out (96 bytes, device); the main stream has joined the side stream when this returns.
static int host_msm_upload(zkhip_ctx* ctx, const uint64_t* scalars, size_t n, size_t K, void* d_s) {
    const size_t per = ((n + K - 1) / K + 255) & ~(size_t)255;
    for (size_t j = 0; j < K; ++j) {
        const size_t off = std::min(n, j * per), len = std::min(per, n - off);
        if (len) ZK_HIP(hipMemcpyAsync((char*)d_s + off * 32, (const char*)scalars + off * 32, len * 32, hipMemcpyHostToDevice, ctx->copy_stream));
        ZK_HIP(hipEventRecord(ctx->host_chunk_event[j], ctx->copy_stream));
    }
    return ZKHIP_OK;
}
// the K chunk copies of one host column as jobs of a copy worker; chunk j's event is host_chunk_event[event_base + j]
static void host_msm_jobs(zkhip_ctx* ctx, const void* host, size_t n, size_t K, void* d_col, size_t event_base, std::vector<zk_copy_job>* jobs) {
    const size_t per = ((n + K - 1) / K + 255) & ~(size_t)255;
    for (size_t j = 0; j < K; ++j) {
        const size_t off = std::min(n, j * per), len = std::min(per, n - off);
        jobs->push_back(zk_copy_job{(char*)d_col + off * 32, (const char*)host + off * 32, len * 32, ctx->host_chunk_event[event_base + j]});
    }
}
static int host_msm_commit(zkhip_ctx* ctx, hipStream_t main, const zkhip_srs* srs, size_t n, size_t K, const void* d_s, void* d_out, zk_copy_worker* worker = nullptr,
                           size_t first_job = 0, size_t event_base = 0) {
    const zkhip_srs* one_srs[1] = {srs};
    const void* cols[1] = {d_s};
    const size_t per = ((n + K - 1) / K + 255) & ~(size_t)255;
    std::vector<MsmPartials> parts(K);
    bool any = false;
    for (size_t j = 0; j < K; ++j) {
        const size_t off = std::min(n, j * per), len = std::min(per, n - off);
        hipStream_t sj = (j & 1) ? ctx->side_stream : main;
        if (worker) ZK_HIP(worker->wait(first_job + j));      // the chunk's event exists only once the copy thread has recorded it
        ZK_HIP(hipStreamWaitEvent(sj, ctx->host_chunk_event[event_base + j], 0));
        ctx->stream = sj;
        char tag[16];
        snprintf(tag, sizeof tag, "#c%zu", j);
        ctx->scratch_tag = tag;      // chunk j's sort / accumulation buffers are its own: they stay alive until the merge below has read them
        ZK_TRY(msm_partials(ctx, one_srs, cols, 1, off, len, &parts[j]));
        any |= !parts[j].empty;
    }
    ctx->scratch_tag.clear();
    ctx->stream = main;
    ZK_HIP(hipEventRecord(ctx->side_event, ctx->side_stream));
    ZK_HIP(hipStreamWaitEvent(main, ctx->side_event, 0));
    if (!any) {
        hipLaunchKernelGGL(k_set_identity, dim3(1), dim3(64), 0, main, (uint32_t*)d_out, 1u);
    } else {
        const uint32_t B = srs->B;
        void *d_mp, *d_mcnt, *d_moff;
        ZK_TRY(ctx->get_scratch("msm_merge_p", (size_t)B * PART_WORDS * 4, &d_mp));
        ZK_TRY(ctx->get_scratch("msm_merge_cnt", (size_t)B * 4, &d_mcnt));
        ZK_TRY(ctx->get_scratch("msm_merge_off", ((size_t)B + 4) * 4, &d_moff));
        MergeArgs A;
        memset(&A, 0, sizeof A);
        A.K = (uint32_t)K;
        for (size_t j = 0; j < K; ++j) if (!parts[j].empty) A.s[j] = MergeSrc{parts[j].p, parts[j].cnt, parts[j].off};
        { ProfScope ps(ctx, "msm_accum_jac");
        hipLaunchKernelGGL(k_merge_buckets, dim3(div_up((size_t)B + 1, 256)), dim3(256), 0, main, A, B, (uint32_t*)d_mp, (uint32_t*)d_mcnt, (uint32_t*)d_moff); }
        ZK_TRY(msm_tail(ctx, 1, MsmPartials{(const uint32_t*)d_mp, (const uint32_t*)d_mcnt, (const uint32_t*)d_moff, B, B, false}, d_out));
    }
    ZK_LAUNCH_CHECK();
    // the side stream's next chunk (of a following column) must not overwrite scratch the merge on the main stream is still reading
    ZK_HIP(hipEventRecord(ctx->side_event, main));
    ZK_HIP(hipStreamWaitEvent(ctx->side_stream, ctx->side_event, 0));
    return ZKHIP_OK;
}
static int host_msm_pipelined(zkhip_ctx* ctx, hipStream_t main, const zkhip_srs* srs, const uint64_t* scalars, size_t n, size_t K, void* d_s, void* d_out) {
    ZK_TRY(host_msm_upload(ctx, scalars, n, K, d_s));
    return host_msm_commit(ctx, main, srs, n, K, d_s, d_out);
}
}  // extern "C"
namespace zk {
// zkhip_create_proof_ex's use of the pipeline for ONE host column that the proof's first commitment needs (the caller's random polynomial): the chunk uploads are issued
// early, ahead of every other upload on the copy stream (host_column_upload; the caller registers the slice and fences the copy stream), the commitment later
// (host_column_commit, on the context's current stream and its side stream; d_out: 96 bytes, device or pinned host).  -> K = 1: not worth it, the caller's plain path.
size_t host_column_chunks(const zkhip_ctx* ctx, const zkhip_srs* srs, size_t n) { return host_msm_chunks(ctx, srs, n); }
int host_column_jobs(zkhip_ctx* ctx, const void* host, size_t n, size_t K, void* d_col, std::vector<zk_copy_job>* jobs) {
    ZK_TRY(host_msm_streams(ctx, K + 1));
    host_msm_jobs(ctx, host, n, K, d_col, 0, jobs);
    return ZKHIP_OK;
}
int host_column_commit(zkhip_ctx* ctx, const zkhip_srs* srs, size_t n, size_t K, const void* d_col, void* d_out, zk_copy_worker* worker, size_t first_job) {
    hipStream_t main = ctx->stream;
    struct Back { zkhip_ctx* c; hipStream_t s; ~Back() { c->stream = s; c->scratch_tag.clear(); } } back{ctx, main};
    return host_msm_commit(ctx, main, srs, n, K, d_col, d_out, worker, first_job);
}
}  // namespace zk
extern "C" {

// best_multiexp on a caller's host slice (the `curves` patch level: halo2curves::msm::best_multiexp -> this, reached from
// /root/reference/src/helpers.rs:233,299 and src/bin/cli.rs:320,369,519 through ParamsKZG::commit / commit_lagrange).
// Small inputs: one upload, one MSM.  From 2^20 scalars (32 MiB: the upload is no longer noise against the sum) the call is PIPELINED:
//   * the K chunk uploads are issued on the copy stream by a worker thread (zk_copy_worker, option host_copy_thread, default): a pageable source makes every
//     hipMemcpyAsync block the thread that issues it for the copy's own duration, and issued from THIS thread the chunked form only adds overhead (measured: 8.2 -> 9.6 ms
//     at 2^22, K = 4).  Pageable memory moves at the pinned rate here (56.5 vs 57.4 GB/s).  host_copy_thread = 0: the slice is registered (hipHostRegister) for the call
//     instead and the copies come from this thread, asynchronous — cheap for pages that were pinned recently (2 us), ~0.4 ms per call for cold ones;
//   * chunk j's digits, sort and bucket accumulation over points [off_j, off_j + len_j) start as soon as ITS bytes have landed, on the main and the
//     side stream alternately, each chunk in its own scratch (scratch_tag);
//   * the K chunks' bucket sums are folded bucket by bucket (k_merge_buckets) and ONE tail finishes: a chunk costs its share of the throughput-bound
//     work plus a plan, not a bucket reduction of its own (K separate MSMs added on the host: 7.2 ms at 2^22 from pinned memory; merged: 6.7 from pageable).
int zkhip_msm_g1(zkhip_ctx* ctx, const zkhip_srs* srs, const uint64_t* scalars, size_t n, uint64_t out_xyz[12]) {
    if (!ctx || !srs || !out_xyz || (!scalars && n)) { set_error("zkhip_msm_g1: null argument"); return ZKHIP_EINVAL; }
    if (ctx->dead || ctx->comm.stuck) { set_error("zkhip_msm_g1: the context was given up on by an earlier host wait (comm_timeout_ms)"); return ZKHIP_EHIP; }
    void *d_s, *d_o;
    ZK_TRY(ctx->get_scratch("msm_host_scalars", (n ? n : 1) * 32, &d_s));
    ZK_TRY(ctx->get_scratch("msm_host_out", 96, &d_o));
    const size_t K = host_msm_chunks(ctx, srs, n);
    uint64_t jac[12];
    if (K == 1) {
        const zkhip_srs* one_srs[1] = {srs};
        const void* cols[1] = {d_s};
        if (n) ZK_HIP(hipMemcpyAsync(d_s, scalars, n * 32, hipMemcpyHostToDevice, ctx->stream));
        ZK_TRY(msm_run(ctx, one_srs, cols, 1, 0, n, d_o));
        ZK_HIP(hipMemcpyAsync(jac, d_o, 96, hipMemcpyDeviceToHost, ctx->stream));
        ZK_HIP(stream_wait(ctx, ctx->stream));      // (polling: a 2^17 MSM is 0.5 ms, an interrupt-driven wait wakes tens of microseconds late)
    } else {
        hipStream_t main = ctx->stream;
        ZK_TRY(host_msm_streams(ctx, K + 1));
        HostMsmRestore restore{ctx, main};
        ZK_TRY(host_msm_fence(ctx, main, ctx->host_chunk_event[K]));
        if (ctx->opt.host_copy_thread != 0) {
            // the chunk copies from a worker thread (common.hpp zk_copy_worker): a pageable slice blocks THAT thread copy by copy, this one launches the chunks' MSMs behind the
            // events as the worker records them.  No registration: pinning pages that were not pinned recently costs ~0.4 ms per call for one slice (0.7 ms each for many)
            zk_copy_worker worker;      // (declared after `restore`: joined — its destructor — before the restore's error path drains the copy stream)
            ZK_TRY(zk::host_column_jobs(ctx, scalars, n, K, d_s, &worker.jobs));
            if (!worker.start(ctx->device, ctx->copy_stream)) ZK_HIP(worker.run_inline(ctx->copy_stream));
            ZK_TRY(host_msm_commit(ctx, main, srs, n, K, d_s, d_o, &worker, 0));
            ZK_HIP(hipMemcpyAsync(jac, d_o, 96, hipMemcpyDeviceToHost, main));
            ZK_HIP(stream_wait(ctx, main));
        } else {
            restore.pin(scalars, n * 32);
            ZK_TRY(host_msm_pipelined(ctx, main, srs, scalars, n, K, d_s, d_o));
            // every chunk's bytes have been consumed once the main stream drains: wait here, so that the slice can be unregistered / dropped
            ZK_HIP(hipMemcpyAsync(jac, d_o, 96, hipMemcpyDeviceToHost, main));
            ZK_HIP(stream_wait(ctx, main));
        }
        restore.ok = true;
    }
    // normalise: (x, y, 1) or the identity (0, 1, 0), like G1::from(G1Affine)
    g1j_store_abi(out_xyz, g1j_from_affine(g1j_to_affine(g1j_load_abi(jac))));
    return ZKHIP_OK;
}

// ncols host columns in one call (SURVEY.md 8(b)'s zkhip_msm_g1_batch: what a patched ParamsKZG would call for a batch of commitments): out_xyz receives
// ncols x 12 u64, each normalised like zkhip_msm_g1's.  Every column goes through the chunk pipeline above, one after the other WITHOUT a host synchronisation in
// between: column j + 1's first chunk is on the wire while column j's last accumulation, merge and tail run, so only the very first chunk's upload is exposed
// (a loop over zkhip_msm_g1 pays it, a read-back and a drained GPU per column).  One wait and one read-back at the end.
int zkhip_msm_g1_batch(zkhip_ctx* ctx, const zkhip_srs* srs, const uint64_t* const* scalar_cols, size_t ncols, size_t n, uint64_t* out_xyz) {
    if (!ctx || !srs || (ncols && (!scalar_cols || !out_xyz))) { set_error("zkhip_msm_g1_batch: null argument"); return ZKHIP_EINVAL; }
    if (ncols == 0) return ZKHIP_OK;
    for (size_t j = 0; j < ncols; ++j) if (!scalar_cols[j] && n) { set_error("zkhip_msm_g1_batch: column %zu is null", j); return ZKHIP_EINVAL; }
    if (n > srs->n_total) { set_error("zkhip_msm_g1_batch: %zu scalars for an SRS of %zu bases", n, srs->n_total); return ZKHIP_EINVAL; }
    if (ctx->dead || ctx->comm.stuck) { set_error("zkhip_msm_g1_batch: the context was given up on by an earlier host wait (comm_timeout_ms)"); return ZKHIP_EHIP; }
    if (n < 65536 || srs->n_total != srs->n || ctx->comm.nranks > 1 || n > srs->n) {      // small or collective: column by column through the plain form
        for (size_t j = 0; j < ncols; ++j) ZK_TRY(zkhip_msm_g1(ctx, srs, scalar_cols[j], n, out_xyz + 12 * j));
        return ZKHIP_OK;
    }
    const size_t K = host_msm_chunks(ctx, srs, n);
    void *d_s, *d_o;
    ZK_TRY(ctx->get_scratch("msm_host_batch_scalars", ncols * n * 32, &d_s));
    ZK_TRY(ctx->get_scratch("msm_host_batch_out", ncols * 96, &d_o));
    hipStream_t main = ctx->stream;
    ZK_TRY(host_msm_streams(ctx, ncols * K + 1));      // an event per (column, chunk): the worker runs ahead of the commits
    HostMsmRestore restore{ctx, main};
    ZK_TRY(host_msm_fence(ctx, main, ctx->host_chunk_event[ncols * K]));
    zk_copy_worker worker;      // (after `restore`: joined before its error path drains the copy stream)
    if (ctx->opt.host_copy_thread != 0) {
        for (size_t j = 0; j < ncols; ++j) host_msm_jobs(ctx, scalar_cols[j], n, K, (char*)d_s + j * n * 32, j * K, &worker.jobs);
        if (!worker.start(ctx->device, ctx->copy_stream)) ZK_HIP(worker.run_inline(ctx->copy_stream));
        for (size_t j = 0; j < ncols; ++j)
            ZK_TRY(host_msm_commit(ctx, main, srs, n, K, (char*)d_s + j * n * 32, (char*)d_o + j * 96, &worker, j * K, j * K));
    } else {
        for (size_t j = 0; j < ncols; ++j) restore.pin(scalar_cols[j], n * 32);
        for (size_t j = 0; j < ncols; ++j)
            ZK_TRY(host_msm_pipelined(ctx, main, srs, scalar_cols[j], n, K, (char*)d_s + j * n * 32, (char*)d_o + j * 96));
    }
    std::vector<uint64_t> jac(12 * ncols);
    ZK_HIP(hipMemcpyAsync(jac.data(), d_o, ncols * 96, hipMemcpyDeviceToHost, main));
    ZK_HIP(stream_wait(ctx, main));
    restore.ok = true;
    for (size_t j = 0; j < ncols; ++j) g1j_store_abi(out_xyz + 12 * j, g1j_from_affine(g1j_to_affine(g1j_load_abi(jac.data() + 12 * j))));
    return ZKHIP_OK;
}

}  // extern "C"
