// ntt.hip — radix-2 NTT / iNTT over BN254 Fr and the EvaluationDomain basis changes, for gfx950.
//
// Drop-in for halo2curves 0.4.0 fft::best_fft (natural order in, natural order out) and for
// halo2_proofs poly::EvaluationDomain::{lagrange_to_coeff, coeff_to_extended, extended_to_coeff,
// divide_by_vanishing_poly} [UPSTREAM-RECALL; crates pinned at /root/reference/Cargo.lock:1359-1361
// and :1320-1322].  Results are the unique DFT values, so the decomposition is free:
//
//   n = n_1 * n_2 * ... * n_p  (p <= 4 passes, each n_q = 2^s_q <= 2^11)
//   pass q < p : for fixed outer digits, a size-n_q DFT over digit q held in LDS, then the
//                Cooley-Tukey twiddle w_n^(prefix * lo * k_q); tiles of T adjacent `lo` values so
//                every global access is a T*32-byte contiguous run (in place).
//   pass p     : contiguous rows of n_p, T rows adjacent in k_1 so the digit-reversed scatter of
//                the outputs is also T*32-byte contiguous (out of place).
//   The zeta coset scaling / zero padding of coeff_to_extended is fused into the first load, the
//   1/n and inverse coset scaling of ifft / extended_to_coeff into the last store.
//
// Bound: integer multiply (one Fr product per butterfly + one per element per non-final pass);
// HBM traffic is 64 B per element per pass.
#include <algorithm>
#include <mutex>

#include "common.hpp"
using namespace zk;

#define NTT_TILE 2048  // elements per workgroup tile, 256 threads; 9 x u32 per element = 72 KiB of LDS

// Twiddle / scale constants live in the library's R' = 2^261 form so that mul(v, w) = v * w for data in
// ANY fixed Montgomery scaling: the NTT is linear, so ABI data (x 2^256) is transformed as it is —
// no domain conversion on load or store, only the limb split / canonical pack.
//
// Lazy bounds inside a tile: a butterfly maps (a, b) -> (a + b w, a - b w + 3p) with b w < 2p, so a
// value grows by at most 3p per stage: <= (2 + 3*11) p = 35 p after an 11-stage tile (limit 120 p).

// ------------------------------------------------------------------ twiddle tables
// table[i] = base^i for i < count, CHK consecutive powers per thread
__global__ void k_powers(uint32_t* table, size_t count, fe base_v) {
    const size_t CHK = 16;
    size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t lo = t * CHK;
    if (lo >= count) return;
    el2<Fr> base(base_v);
    el2<Fr> w = pow_u64<Fr>(base, (uint64_t)lo);
    size_t hi = lo + CHK < count ? lo + CHK : count;
    for (size_t i = lo; i < hi; ++i) {
        store_raw<Fr>(table + i * 8, w);
        w = w * base;
    }
}

int zkhip_ctx::get_twiddles(const uint64_t omega[4], uint32_t log_n, const Twiddle** out) {
    for (auto& t : twiddles)
        if (t.log_n == log_n && memcmp(t.omega, omega, 32) == 0) { *out = &t; return ZKHIP_OK; }
    Twiddle t;
    t.log_n = log_n;
    memcpy(t.omega, omega, 32);
    t.h = (log_n + 1) / 2;
    t.bf_bits = log_n >= 1 ? std::min<uint32_t>(10, log_n - 1) : 0;
    size_t n_lo = (size_t)1 << t.h, n_hi = (size_t)1 << (log_n - t.h), n_bf = (size_t)1 << t.bf_bits;
    void* base;
    hipError_t e = zk::dev_malloc((void**)&base, (n_lo + n_hi + n_bf) * 32);
    if (e != hipSuccess) { (void)hipGetLastError(); set_error("hipMalloc twiddles: %s", hipGetErrorString(e)); return ZKHIP_ENOMEM; }
    t.d_lo = base;
    t.d_hi = (char*)base + n_lo * 32;
    t.d_bf = (char*)base + (n_lo + n_hi) * 32;
    el2<Fr> w = from_abi<Fr>(mem_load(omega));
    el2<Fr> w_hi = pow_u64<Fr>(w, (uint64_t)1 << t.h);
    el2<Fr> w_bf = log_n >= 1 ? pow_u64<Fr>(w, (uint64_t)1 << (log_n - 1 - t.bf_bits)) : w;
    auto launch = [&](void* tab, size_t cnt, const el2<Fr>& b) {
        hipLaunchKernelGGL(k_powers, dim3(div_up(div_up(cnt, 16), 64)), dim3(64), 0, stream, (uint32_t*)tab, cnt, b.v);
    };
    launch(t.d_lo, n_lo, w);
    launch(t.d_hi, n_hi, w_hi);
    launch(t.d_bf, n_bf, w_bf);
    ZK_LAUNCH_CHECK();
    ZK_HIP(hipStreamSynchronize(stream));   // one-time: the tables are shared by every stream of the context
    twiddles.push_back(t);
    *out = &twiddles.back();
    return ZKHIP_OK;
}

struct TwDev {   // device view
    const uint32_t* lo; const uint32_t* hi; const uint32_t* bf;
    uint32_t h, bf_shift;   // bf index of w^(j << (m-1-st)) is j << (bf_bits - st) ... see tile_ntt
};

// ------------------------------------------------------------------ kernels
struct NttScale {
    fe pre[3];   // element i is multiplied by pre[i % 3] on the first load (if use_pre); R' form, < 2p
    fe post[3];  // output k is multiplied by post[k % 3] on the last store (if use_post)
    int use_pre, use_post;
    // per-index tables (raw R' form), or null: element i is multiplied by pre_tab[i] on the first load, output k by post_tab[k] on
    // the last store — the coset shifts s^i / s^-k of the quotient's coset-by-coset evaluation (cosets.hip)
    const uint32_t* pre_tab;
    const uint32_t* post_tab;
    // polynomial p of the batch uses table (p / tab_group): tables are tab_stride elements apart (tab_p0: the launch's first polynomial)
    uint32_t tab_group, tab_stride, tab_p0;
};
__device__ __forceinline__ const uint32_t* tab_of(const uint32_t* tab, const NttScale& sc, uint32_t p) {
    return tab ? tab + (size_t)((sc.tab_p0 + p) / sc.tab_group) * sc.tab_stride * 8 : nullptr;
}
using tile_el = el<Fr, 40 * U>;   // anything held in a tile

extern __shared__ uint32_t ntt_lds[];   // fe tile[NTT_TILE]

__device__ __forceinline__ uint32_t bitrev(uint32_t v, uint32_t bits) { return bits ? (__brev(v) >> (32 - bits)) : 0; }

__device__ __forceinline__ el2<Fr> twiddle_at(const TwDev& tw, uint32_t e) {
    // w^e = lo[e mod 2^h] * hi[e >> h]: two cache-resident loads and one product
    return load_raw<Fr>(tw.lo + (size_t)(e & ((1u << tw.h) - 1)) * 8) * load_raw<Fr>(tw.hi + (size_t)(e >> tw.h) * 8);
}

// s radix-2 DIT stages on a tile laid out tile[row * T + tl], rows = 2^s (rows were loaded bit-reversed).
__device__ __forceinline__ void tile_ntt(fe* tile, uint32_t s, uint32_t logT, const TwDev& tw) {
    uint32_t T = 1u << logT;
    uint32_t nbf = (1u << s) * T / 2;
    for (uint32_t st = 0; st < s; ++st) {
        __syncthreads();
        uint32_t half = 1u << st;
        for (uint32_t bf = threadIdx.x; bf < nbf; bf += blockDim.x) {
            uint32_t tl = bf & (T - 1), pr = bf >> logT;
            uint32_t j = pr & (half - 1);
            uint32_t r0 = ((pr >> st) << (st + 1)) | j;
            uint32_t i0 = r0 * T + tl, i1 = (r0 + half) * T + tl;
            fe a = tile[i0];
            el2<Fr> bw;
            // w^(j << (m-1-st)) = bf[j << (bf_bits - st)]   (tw.bf_shift = bf_bits; s <= bf_bits + 1)
            if (j != 0) bw = tile_el(tile[i1]) * load_raw<Fr>(tw.bf + ((size_t)j << (tw.bf_shift - st)) * 8);
            else if (st == 0) bw = el2<Fr>(tile[i1]);   // fresh loads are < 2p: twiddle 1 needs no product
            else bw = canonical(tile_el(tile[i1]));      // twiddle 1 later on: contract (conditional subtractions, no product) so the +3p/stage bound holds
            fe sum, dif;
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                sum.l[q] = a.l[q] + bw.v.l[q];
                dif.l[q] = a.l[q] + kp_spread<Fr>(3, q) - bw.v.l[q];
            }
            fe_normalize(sum);
            fe_normalize(dif);
            tile[i0] = sum;
            tile[i1] = dif;
        }
    }
    __syncthreads();
}

__device__ __forceinline__ fe load_in(const uint32_t* src, uint32_t i, uint32_t n_in, const NttScale& sc, const uint32_t* pre_tab) {
    if (i >= n_in) return fe_zero();
    el1<Fr> v = load_raw<Fr>(src + (size_t)i * 8);
    if (sc.use_pre) {
        uint32_t r = i % 3;
        if (r == 1) return (v * el2<Fr>(sc.pre[1])).v;
        if (r == 2) return (v * el2<Fr>(sc.pre[2])).v;
    }
    if (pre_tab) return (v * load_raw<Fr>(pre_tab + (size_t)i * 8)).v;
    return v.v;
}

// Non-final pass: position = (hi << (s + lo_bits)) | (digit << lo_bits) | lo.
// One workgroup transforms the same tile position of `group` polynomials one after the other: the Cooley-Tukey twiddles
// w^(lo * r << hi_bits) of its 8 outputs per thread depend on the position only, so they are built once (two table loads and a
// product each) and reused — a quarter of the pass's products when every polynomial pays for them.
__global__ void __launch_bounds__(256) k_ntt_strided(const uint32_t* const* srcs, uint32_t* const* dsts, uint32_t npolys, uint32_t group,
                                                      uint32_t m, uint32_t s, uint32_t lo_bits, uint32_t logT, uint32_t n_in, TwDev tw,
                                                      NttScale sc) {
    fe* tile = reinterpret_cast<fe*>(ntt_lds);
    uint32_t T = 1u << logT, rows = 1u << s;
    uint32_t hi_bits = m - s - lo_bits;
    uint32_t tiles_lo = 1u << (lo_bits - logT);
    uint32_t lo_tile = blockIdx.x & (tiles_lo - 1), hi = blockIdx.x >> (lo_bits - logT);
    uint32_t lo0 = lo_tile << logT;
    uint32_t base = (hi << (s + lo_bits)) | lo0;
    uint32_t cnt = rows * T;
    fe twr[NTT_TILE / 256];   // this thread's output twiddles (the exponent-0 ones are the field's one: a product is cheaper than a branch here)
#pragma clang loop unroll(full)
    for (uint32_t it = 0; it < NTT_TILE / 256; ++it) {
        uint32_t e = threadIdx.x + it * 256;
        uint32_t tl = e & (T - 1), r = e >> logT;
        twr[it] = (e < cnt ? twiddle_at(tw, ((lo0 | tl) * r) << hi_bits) : el2<Fr>(one<Fr>())).v;
    }
    const uint32_t p0 = blockIdx.y * group, p1 = min(npolys, p0 + group);
    for (uint32_t pi = p0; pi < p1; ++pi) {
        const uint32_t* src = srcs[pi];
        uint32_t* dst = dsts[pi];
        for (uint32_t e = threadIdx.x; e < cnt; e += blockDim.x) {
            uint32_t tl = e & (T - 1), j = e >> logT;
            uint32_t pos = base | (j << lo_bits) | tl;
            tile[bitrev(j, s) * T + tl] = load_in(src, pos, n_in, sc, tab_of(sc.pre_tab, sc, pi));
        }
        tile_ntt(tile, s, logT, tw);
#pragma clang loop unroll(full)
        for (uint32_t it = 0; it < NTT_TILE / 256; ++it) {
            uint32_t e = threadIdx.x + it * 256;
            if (e < cnt) {
                uint32_t tl = e & (T - 1), r = e >> logT;
                tile_el v(tile[r * T + tl]);
                store_packed<Fr>(dst + (size_t)(base | (r << lo_bits) | tl) * 8, v * el2<Fr>(twr[it]));   // < 2p, read by the next pass only
            }
        }
        __syncthreads();   // the tile is reloaded for the next polynomial
    }
}

// Final pass: rows of 2^s contiguous elements; T rows adjacent in the first digit k_1.
// digit widths of the earlier passes are in sw[0..np-2] (sw[0] = s_1 is the most significant slot).
struct NttDigits { uint32_t np; uint32_t sw[6]; };

__global__ void __launch_bounds__(256) k_ntt_final(const uint32_t* const* srcs, uint32_t* const* dsts, uint32_t m, uint32_t s,
                                                    uint32_t logT, uint32_t n_in, TwDev tw, NttScale sc, NttDigits dg) {
    fe* tile = reinterpret_cast<fe*>(ntt_lds);
    const uint32_t* src = srcs[blockIdx.y];
    uint32_t* dst = dsts[blockIdx.y];
    uint32_t T = 1u << logT, rows = 1u << s;
    uint32_t hi_bits = m - s;                 // bits of the row index
    uint32_t s1 = dg.np > 1 ? dg.sw[0] : 0;   // k_1 is the top s1 bits of the row index
    uint32_t rest_bits = hi_bits - s1;
    // blockIdx.x -> (k1 tile, rest)
    uint32_t rest = blockIdx.x & ((1u << rest_bits) - 1);
    uint32_t k1_0 = (blockIdx.x >> rest_bits) << logT;
    uint32_t cnt = rows * T;
    for (uint32_t e = threadIdx.x; e < cnt; e += blockDim.x) {
        uint32_t j = e & (rows - 1), tl = e >> s;
        uint32_t row = ((k1_0 + tl) << rest_bits) | rest;
        tile[bitrev(j, s) * T + tl] = load_in(src, (row << s) | j, n_in, sc, tab_of(sc.pre_tab, sc, blockIdx.y));
    }
    tile_ntt(tile, s, logT, tw);
    // output index: k = k_1 + k_2 2^{s_1} + ... ; digits k_2..k_{p-1} come out of `rest` (slot order, msb first)
    uint32_t kbase = 0, shift_out = s1, rem = rest, rb = rest_bits;
    for (uint32_t q = 1; q + 1 < dg.np; ++q) {
        uint32_t w = dg.sw[q];
        rb -= w;
        uint32_t d = (rem >> rb) & ((1u << w) - 1);
        kbase |= d << shift_out;
        shift_out += w;
    }
    for (uint32_t e = threadIdx.x; e < cnt; e += blockDim.x) {
        uint32_t tl = e & (T - 1), r = e >> logT;
        uint32_t k = (r << hi_bits) | kbase | (k1_0 + tl);
        tile_el v(tile[r * T + tl]);
        void* out = dst + (size_t)k * 8;
        if (sc.use_post) store_raw<Fr>(out, v * el2<Fr>(sc.post[k % 3]));
        else if (sc.post_tab) store_raw<Fr>(out, v * load_raw<Fr>(tab_of(sc.post_tab, sc, blockIdx.y) + (size_t)k * 8));
        else store_raw<Fr>(out, v);
    }
}

// ------------------------------------------------------------------ register-tiled variants (full 2048-element tiles)
// The same tile transform with three radix-2 stages at a time held in registers: thread t owns 8 elements whose row indices
// differ in one 3-bit field ("slot"), runs the stages whose partner bit lies in that field, and only then exchanges through
// LDS — ceil(s/3) - 1 round trips and barriers per tile instead of s, and the first load / last store go straight between
// HBM and registers.  Group i covers stages [3i, e_i), e_i = min(3i+3, s); its slot field is the row bits [e_i-3, e_i), so a
// short last group simply carries 3 - g passive bits.  Logical tile index L = row * T + tl as above; the LDS element index is
// L with a few high bits XORed into low ones (Swz) so that the 32 lanes of an access, which differ in bits on both sides of the
// slot field, land in 32 different banks (an element is 9 dwords, 9 is odd).
// Three rules, chosen per launch (compile-time in the kernel: an index costs three instructions and there are 2 E of them per exchange).
//   SWZ_FIELD : the field L[sh ...] & mk XORed in at bit ts — the strided passes (slot of group 0 at logT: sh = logT + SB) and the final
//               pass's tiles with logT + SB >= 5 (its first write runs the lanes along the bit-REVERSED top row bits: sh = log tile - 5).
//   SWZ_FIELD2: the final pass's tiles with logT + SB <= 4, where the slot of the second group still lies inside the five bank bits and the
//               one field folds two lane bits onto one bank bit (2- / 4-way conflicts; measured on the first 4-per-thread build: 6.6 M /
//               23.6 M conflict cycles per dispatch at 2^19 / 2^21): a second field, L[5 ...] of slot width, under the top of the bank bits.
//   SWZ_REV   : SB = 2 at logT = 0, where no two fields do: bank bit j takes the row bits in reversed order (L8, L7, L6, L5 ^ L10, L6 ^ L9).
// tests/test_ntt_swizzle.py replays every access of every tile shape against these rules (32 lanes, 32 banks, 9-dword elements).
struct Swz { uint32_t sh, mk, ts; };
enum { SWZ_FIELD = 0, SWZ_FIELD2 = 1, SWZ_REV = 2 };
template <int SB, int RULE>
__device__ __forceinline__ uint32_t swz(uint32_t L, const Swz& z) {
    if constexpr (RULE == SWZ_REV) {
        const uint32_t x = L >> 5;
        return L ^ (__brev(x & 15u) >> 28) ^ (((x >> 5) & 1u) << 3) ^ ((((x >> 1) ^ (x >> 4)) & 1u) << 4);
    } else {
        uint32_t a = L ^ (((L >> z.sh) & z.mk) << z.ts);
        if constexpr (RULE == SWZ_FIELD2) a ^= ((L >> 5) & ((1u << SB) - 1)) << (5 - SB);
        return a;
    }
}
// thread index -> logical index with a zero SB-bit slot field at bit p.  SB = 3: 8 elements per thread, 256 threads per tile, two waves per
// SIMD (the tile's 72 KiB of LDS allow two workgroups per CU either way); SB = 2: 4 elements per thread, 512 threads, four waves per SIMD
// at <= 128 registers, for two more exchanges per 11-stage tile.
template <int SB>
__device__ __forceinline__ uint32_t place(uint32_t t, uint32_t p) { return (t & ((1u << p) - 1)) | ((t >> p) << (p + SB)); }

// (a, b) <- (a + bw, a - bw + 3p) WITHOUT carry propagation: bw is normalised (limbs < 2^29); a's limbs may be lazy.  Per stage a limb
// grows by < 2^29 (sum) or < 2^30 (difference: the borrow-spread constant), so from normalised inputs three stages leave every limb below
// 7 * 2^29 < 2^32, and a lazy multiplicand of the third stage (limbs < 5 * 2^29) keeps the product's 64-bit columns below
// 9 * 5 * 2^58 + 9 * 2^58 < 2^64.  A register group (<= 3 stages) therefore normalises ONCE, when it hands its values on (norm_all) — a third
// of the carry passes of the stage-by-stage form, 9 % of the kernels' instructions.  The value bound (+3p per stage) is unchanged.
__device__ __forceinline__ void bfly(fe& a, fe& b, const el2<Fr>& bw) {
    fe sum, dif;
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        sum.l[q] = a.l[q] + bw.v.l[q];
        dif.l[q] = a.l[q] + kp_spread<Fr>(3, q) - bw.v.l[q];
    }
    a = sum;
    b = dif;
}
template <int E>
__device__ __forceinline__ void norm_all(fe (&v)[E]) {
#pragma unroll
    for (int q = 0; q < E; ++q) fe_normalize(v[q]);
}
// a unit twiddle after the first stage: contract the (lazy) operand by conditional subtractions instead of a product by one
__device__ __forceinline__ el1<Fr> unit_operand(fe x) {
    fe_normalize(x);
    return canonical(tile_el(x));
}

// stages 0..SB-1 on freshly loaded values (< 2p); slot = row bits 0..SB-1, so the twiddle exponents are compile-time:
// stage u pairs (q, q | 1 << u) with w_{2^(u+1)}^(q mod 2^u).  SB = 3: 5 products for 12 butterflies; SB = 2: 1 for 4.
template <int SB>
__device__ __forceinline__ void stages_first(fe (&v)[1 << SB], const TwDev& tw) {
    constexpr int E = 1 << SB;
#pragma unroll
    for (int q = 0; q < E; q += 2) bfly(v[q], v[q + 1], el2<Fr>(v[q + 1]));
    el2<Fr> w4 = load_raw<Fr>(tw.bf + ((size_t)1 << (tw.bf_shift - 1)) * 8);
#pragma unroll
    for (int h = 0; h < E; h += 4) {
        bfly(v[h], v[h + 2], unit_operand(v[h + 2]));
        bfly(v[h + 1], v[h + 3], tile_el(v[h + 3]) * w4);
    }
    if constexpr (SB == 3) {
        bfly(v[0], v[4], unit_operand(v[4]));
        bfly(v[1], v[5], tile_el(v[5]) * load_raw<Fr>(tw.bf + ((size_t)1 << (tw.bf_shift - 2)) * 8));
        bfly(v[2], v[6], tile_el(v[6]) * w4);
        bfly(v[3], v[7], tile_el(v[7]) * load_raw<Fr>(tw.bf + ((size_t)3 << (tw.bf_shift - 2)) * 8));
    }
    norm_all(v);
}

// stages [e - G_, e) with the slot at row bits [e-SB, e); rho0 = this thread's row index with a zero slot field.
// The twiddle of a pair depends on the row bits below the partner bit only: 2^(SB-G_+u) distinct ones in stage u.
template <int SB, int G_, int U_>
__device__ __forceinline__ void stage_general(fe (&v)[1 << SB], uint32_t rho0, uint32_t e, const TwDev& tw) {
    constexpr int pb = SB - G_ + U_;   // partner bit inside the slot
    const uint32_t st = e - G_ + U_;
    const uint32_t msk = (1u << st) - 1;
#pragma unroll
    for (int lb = 0; lb < (1 << pb); ++lb) {
        uint32_t j = (rho0 | ((uint32_t)lb << (e - SB))) & msk;
        el2<Fr> w = load_raw<Fr>(tw.bf + ((size_t)j << (tw.bf_shift - st)) * 8);
#pragma unroll
        for (int hb = 0; hb < ((1 << (SB - 1)) >> pb); ++hb) {
            const int q0 = lb | (hb << (pb + 1)), q1 = q0 | (1 << pb);
            bfly(v[q0], v[q1], tile_el(v[q1]) * w);
        }
    }
}
template <int SB, int G_>
__device__ __forceinline__ void stages_general(fe (&v)[1 << SB], uint32_t rho0, uint32_t e, const TwDev& tw) {
    stage_general<SB, G_, 0>(v, rho0, e, tw);
    if constexpr (G_ > 1) stage_general<SB, G_, 1>(v, rho0, e, tw);
    if constexpr (G_ > 2) stage_general<SB, G_, 2>(v, rho0, e, tw);
    norm_all(v);
}

// groups 1.. of a tile whose group 0 is already done in v (slot at logical bit logT, held by the thread that place() would
// number t_first); leaves v in the last group's ownership (slot at logical bit logT + s - SB) and returns that thread's
// logical base index.
template <int SB, int RULE>
__device__ __forceinline__ uint32_t tile_rest(fe (&v)[1 << SB], fe* tile, uint32_t s, uint32_t logT, const Swz& z, const TwDev& tw,
                                              uint32_t t_first) {
    constexpr int E = 1 << SB;
    const uint32_t t = threadIdx.x;
    uint32_t p_prev = logT;
    const uint32_t G = (s + SB - 1) / SB;
#pragma clang loop unroll(disable)
    for (uint32_t i = 1; i < G; ++i) {
        uint32_t e = min(SB * i + SB, s), g = e - SB * i, p = logT + e - SB;
        uint32_t Lw = place<SB>(i == 1 ? t_first : t, p_prev);
#pragma unroll
        for (int q = 0; q < E; ++q) tile[swz<SB, RULE>(Lw | ((uint32_t)q << p_prev), z)] = v[q];
        __syncthreads();
        uint32_t Lr = place<SB>(t, p);
#pragma unroll
        for (int q = 0; q < E; ++q) v[q] = tile[swz<SB, RULE>(Lr | ((uint32_t)q << p), z)];
        uint32_t rho0 = Lr >> logT;
        if constexpr (SB == 3) {
            if (g == 3) stages_general<3, 3>(v, rho0, e, tw);
            else if (g == 2) stages_general<3, 2>(v, rho0, e, tw);
            else stages_general<3, 1>(v, rho0, e, tw);
        } else {
            if (g == 2) stages_general<2, 2>(v, rho0, e, tw);
            else stages_general<2, 1>(v, rho0, e, tw);
        }
        p_prev = p;
    }
    return place<SB>(G == 1 ? t_first : t, p_prev);
}

// table[pos] = w^((lo * r) << hi_bits) for pos = (r << lo_bits) | lo: the Cooley-Tukey twiddle of a non-final pass's output at
// tile-relative position pos (independent of the outer digits).  Built once per (omega, log_n, pass) and kept: with 288 GB of
// HBM a 32-byte-per-element table (512 MiB at 2^24) is cheaper than the 72 registers per thread that caching the twiddles of a
// workgroup cost (one wave per SIMD instead of two) or the product per element that rebuilding them costs.
__global__ void k_pass_twiddles(uint32_t* table, uint32_t count, uint32_t lo_bits, uint32_t hi_bits, TwDev tw) {
    uint32_t pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= count) return;
    uint32_t lo = pos & ((1u << lo_bits) - 1), r = pos >> lo_bits;
    store_raw<Fr>(table + (size_t)pos * 8, twiddle_at(tw, (lo * r) << hi_bits));
}

// the polynomials of a launch, by value in the kernel arguments: no pointer table to upload (an upload is a blit kernel on the
// stream, ~7 us of it per call); longer batches go in launches of NTT_MAXP
#define NTT_MAXP 16
struct NttPtrs { const uint32_t* src[NTT_MAXP]; uint32_t* dst[NTT_MAXP]; };
template <int SB>
__device__ __forceinline__ void ntt_strided_rt(const NttPtrs& PT, uint32_t m, uint32_t s, uint32_t lo_bits, uint32_t logT, uint32_t n_in,
                                               const TwDev& tw, const NttScale& sc, const Swz& z, const uint32_t* __restrict__ ptab) {
    constexpr int E = 1 << SB;
    fe* tile = reinterpret_cast<fe*>(ntt_lds);
    const uint32_t T = 1u << logT, t = threadIdx.x;
    uint32_t tiles_lo = 1u << (lo_bits - logT);
    uint32_t lo_tile = blockIdx.x & (tiles_lo - 1), hi = blockIdx.x >> (lo_bits - logT);
    uint32_t lo0 = lo_tile << logT;
    uint32_t base = (hi << (s + lo_bits)) | lo0;
    const uint32_t* src = PT.src[blockIdx.y];
    uint32_t* dst = PT.dst[blockIdx.y];
    const uint32_t* pre_tab = tab_of(sc.pre_tab, sc, blockIdx.y);
    // ownership at the load: slot = row bits 0..SB-1; row rho <-> digit j = bitrev(rho, s)
    const uint32_t tl0 = t & (T - 1), jrest = bitrev(t >> logT, s - SB);
    fe v[E];
#pragma clang loop unroll(full)
    for (int q = 0; q < E; ++q) {
        const uint32_t qr = SB == 3 ? ((q & 1) << 2) | (q & 2) | (q >> 2) : ((q & 1) << 1) | (q >> 1);   // bitrev of the slot
        uint32_t j = (qr << (s - SB)) | jrest;
        v[q] = load_in(src, base | (j << lo_bits) | tl0, n_in, sc, pre_tab);
    }
    stages_first<SB>(v, tw);
    const uint32_t L_last = tile_rest<SB, SWZ_FIELD>(v, tile, s, logT, z, tw, t);
    const uint32_t p_last = logT + s - SB;
#pragma clang loop unroll(full)
    for (int q = 0; q < E; ++q) {
        uint32_t L = L_last | ((uint32_t)q << p_last);
        uint32_t tl = L & (T - 1), r = L >> logT;
        uint32_t rel = (r << lo_bits) | lo0 | tl;
        store_packed<Fr>(dst + ((size_t)(hi << (s + lo_bits)) | rel) * 8, tile_el(v[q]) * load_raw<Fr>(ptab + (size_t)rel * 8));   // < 2p, read by the next pass only
    }
}
__global__ void __launch_bounds__(256, 2) k_ntt_strided_r8(const NttPtrs PT, uint32_t m, uint32_t s,
                                                            uint32_t lo_bits, uint32_t logT, uint32_t n_in, TwDev tw, NttScale sc, Swz z,
                                                            const uint32_t* __restrict__ ptab) {
    ntt_strided_rt<3>(PT, m, s, lo_bits, logT, n_in, tw, sc, z, ptab);
}
__global__ void __launch_bounds__(512, 4) k_ntt_strided_r4(const NttPtrs PT, uint32_t m, uint32_t s,
                                                            uint32_t lo_bits, uint32_t logT, uint32_t n_in, TwDev tw, NttScale sc, Swz z,
                                                            const uint32_t* __restrict__ ptab) {
    ntt_strided_rt<2>(PT, m, s, lo_bits, logT, n_in, tw, sc, z, ptab);
}

__global__ void __launch_bounds__(256, 4) k_ntt_strided_r4s(const NttPtrs PT, uint32_t m, uint32_t s,
                                                             uint32_t lo_bits, uint32_t logT, uint32_t n_in, TwDev tw, NttScale sc, Swz z,
                                                             const uint32_t* __restrict__ ptab) {
    ntt_strided_rt<2>(PT, m, s, lo_bits, logT, n_in, tw, sc, z, ptab);
}

template <int SB, int RULE>
__device__ __forceinline__ void ntt_final_rt(const NttPtrs& PT, uint32_t m, uint32_t s, uint32_t logT, uint32_t n_in, const TwDev& tw,
                                             const NttScale& sc, const NttDigits& dg, const Swz& z) {
    constexpr int E = 1 << SB;
    fe* tile = reinterpret_cast<fe*>(ntt_lds);
    const uint32_t* src = PT.src[blockIdx.y];
    uint32_t* dst = PT.dst[blockIdx.y];
    const uint32_t* pre_tab = tab_of(sc.pre_tab, sc, blockIdx.y);
    const uint32_t* post_tab = tab_of(sc.post_tab, sc, blockIdx.y);
    const uint32_t T = 1u << logT, t = threadIdx.x;
    uint32_t hi_bits = m - s;
    uint32_t s1 = dg.np > 1 ? dg.sw[0] : 0;
    uint32_t rest_bits = hi_bits - s1;
    uint32_t rest = blockIdx.x & ((1u << rest_bits) - 1);
    uint32_t k1_0 = (blockIdx.x >> rest_bits) << logT;
    // load: lanes run along j (contiguous in memory); thread = (jl, tl), its E values j = jl | qq << (s-SB) are the rows
    // rho = bitrev(jl, s-SB) * E + bitrev(qq, SB)
    fe v[E];
    const uint32_t jl = t & ((1u << (s - SB)) - 1), tl0 = t >> (s - SB);
    const uint32_t row0 = ((k1_0 + tl0) << rest_bits) | rest;
#pragma clang loop unroll(full)
    for (int q = 0; q < E; ++q) {
        const uint32_t qr = SB == 3 ? ((q & 1) << 2) | (q & 2) | (q >> 2) : ((q & 1) << 1) | (q >> 1);
        v[q] = load_in(src, (row0 << s) | (qr << (s - SB)) | jl, n_in, sc, pre_tab);
    }
    stages_first<SB>(v, tw);
    // group 0's ownership in the common (tl, rest-of-row) thread numbering
    const uint32_t L_last = tile_rest<SB, RULE>(v, tile, s, logT, z, tw, tl0 | (bitrev(jl, s - SB) << logT));
    uint32_t kbase = 0, shift_out = s1, rb = rest_bits;
    for (uint32_t q = 1; q + 1 < dg.np; ++q) {
        uint32_t w = dg.sw[q];
        rb -= w;
        uint32_t d = (rest >> rb) & ((1u << w) - 1);
        kbase |= d << shift_out;
        shift_out += w;
    }
    const uint32_t p_last = logT + s - SB;
#pragma clang loop unroll(full)
    for (int q = 0; q < E; ++q) {
        uint32_t L = L_last | ((uint32_t)q << p_last);
        uint32_t tl = L & (T - 1), r = L >> logT;
        uint32_t k = (r << hi_bits) | kbase | (k1_0 + tl);
        void* out = dst + (size_t)k * 8;
        if (sc.use_post) store_raw<Fr>(out, tile_el(v[q]) * el2<Fr>(sc.post[k % 3]));
        else if (post_tab) store_raw<Fr>(out, tile_el(v[q]) * load_raw<Fr>(post_tab + (size_t)k * 8));
        else store_raw<Fr>(out, tile_el(v[q]));
    }
}
template <int RULE>
__global__ void __launch_bounds__(256, 2) k_ntt_final_r8(const NttPtrs PT, uint32_t m, uint32_t s,
                                                       uint32_t logT, uint32_t n_in, TwDev tw, NttScale sc, NttDigits dg, Swz z) {
    ntt_final_rt<3, RULE>(PT, m, s, logT, n_in, tw, sc, dg, z);
}
template <int RULE>
__global__ void __launch_bounds__(512, 4) k_ntt_final_r4(const NttPtrs PT, uint32_t m, uint32_t s,
                                                       uint32_t logT, uint32_t n_in, TwDev tw, NttScale sc, NttDigits dg, Swz z) {
    ntt_final_rt<2, RULE>(PT, m, s, logT, n_in, tw, sc, dg, z);
}

template <int RULE>
__global__ void __launch_bounds__(256, 4) k_ntt_final_r4s(const NttPtrs PT, uint32_t m, uint32_t s,
                                                        uint32_t logT, uint32_t n_in, TwDev tw, NttScale sc, NttDigits dg, Swz z) {
    ntt_final_rt<2, RULE>(PT, m, s, logT, n_in, tw, sc, dg, z);
}

__global__ void k_mul_periodic(uint32_t* a, size_t n, const uint32_t* tev, uint32_t period_mask) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    store_raw<Fr>(a + i * 8, load_raw<Fr>(a + i * 8) * load_raw<Fr>(tev + (size_t)(i & period_mask) * 8));
}

// ------------------------------------------------------------------ host driver
static uint32_t ilog2(uint32_t v) { uint32_t l = 0; while ((1u << (l + 1)) <= v) ++l; return l; }

// the per-pass output twiddle table of a non-final pass (see k_pass_twiddles), cached in the context for its lifetime
static int pass_twiddles(zkhip_ctx* ctx, const zkhip_ctx::Twiddle* twh, const TwDev& tw, uint32_t q, uint32_t s, uint32_t lo_bits,
                         const void** out) {
    char name[160];
    snprintf(name, sizeof name, "ntt_pass_tw:%016llx%016llx%016llx%016llx:%u:%u:%u:%u", (unsigned long long)twh->omega[3],
             (unsigned long long)twh->omega[2], (unsigned long long)twh->omega[1], (unsigned long long)twh->omega[0], twh->log_n, q, s, lo_bits);
    auto it = ctx->persistent.find(name);
    if (it != ctx->persistent.end()) { *out = it->second; return ZKHIP_OK; }
    uint32_t count = 1u << (s + lo_bits), hi_bits = twh->log_n - s - lo_bits;
    void* d;
    hipError_t e = zk::dev_malloc((void**)&d, (size_t)count * 32);
    if (e != hipSuccess) { (void)hipGetLastError(); set_error("hipMalloc pass twiddles: %s", hipGetErrorString(e)); return ZKHIP_ENOMEM; }
    hipLaunchKernelGGL(k_pass_twiddles, dim3(div_up((size_t)count, 256)), dim3(256), 0, ctx->stream, (uint32_t*)d, count, lo_bits, hi_bits, tw);
    ZK_LAUNCH_CHECK();
    ZK_HIP(hipStreamSynchronize(ctx->stream));   // one-time: other streams of the context read the table too
    ctx->persistent[name] = d;
    *out = d;
    return ZKHIP_OK;
}

// srcs/dsts: host arrays of device pointers (src[i] may equal dst[i]); src has n_in valid elements.
static int ntt_run(zkhip_ctx* ctx, const void* const* srcs, void* const* dsts, size_t npolys, const uint64_t omega[4],
                   uint32_t m, uint32_t n_in, const NttScale& sc_in) {
    if (npolys == 0) return ZKHIP_OK;
    if (m > 26) { set_error("ntt: log_n = %u unsupported (max 26)", m); return ZKHIP_EINVAL; }
    hipStream_t st = ctx->stream;
    size_t n = (size_t)1 << m;
    void* d_ptrs;
    ZK_TRY(ctx->get_scratch("ntt_ptrs", npolys * sizeof(void*) * 3, &d_ptrs));
    const uint32_t** d_src = (const uint32_t**)d_ptrs;
    uint32_t** d_dst = (uint32_t**)d_ptrs + npolys;
    uint32_t** d_tmp = (uint32_t**)d_ptrs + 2 * npolys;
    if (m == 0) {  // size-1 transform: copy / scale only
        for (size_t i = 0; i < npolys; ++i)
            if (srcs[i] != dsts[i]) ZK_HIP(hipMemcpyAsync(dsts[i], srcs[i], 32, hipMemcpyDeviceToDevice, st));
        return ZKHIP_OK;
    }
    const zkhip_ctx::Twiddle* twh;
    ZK_TRY(ctx->get_twiddles(omega, m, &twh));
    TwDev tw{(const uint32_t*)twh->d_lo, (const uint32_t*)twh->d_hi, (const uint32_t*)twh->d_bf, twh->h, twh->bf_bits};
    static std::once_flag lds_attr_once;   // 72 KiB of dynamic LDS needs the opt-in once per process
    hipError_t attr_err = hipSuccess;
    std::call_once(lds_attr_once, [&] {
        attr_err = hipFuncSetAttribute((const void*)k_ntt_strided, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(NTT_TILE * sizeof(fe)));
        for (const void* f : {(const void*)k_ntt_final, (const void*)k_ntt_strided_r8, (const void*)k_ntt_strided_r4, (const void*)k_ntt_final_r8<SWZ_FIELD>, (const void*)k_ntt_final_r8<SWZ_FIELD2>,
                              (const void*)k_ntt_final_r4<SWZ_FIELD>, (const void*)k_ntt_final_r4<SWZ_FIELD2>, (const void*)k_ntt_final_r4<SWZ_REV>})
            if (attr_err == hipSuccess) attr_err = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(NTT_TILE * sizeof(fe)));
        // the occupancy experiment (ntt_lds_pad) asks for up to a whole CU's LDS per workgroup of the 4-per-thread kernels
        for (const void* f : {(const void*)k_ntt_strided_r4, (const void*)k_ntt_strided_r4s, (const void*)k_ntt_final_r4<SWZ_FIELD>, (const void*)k_ntt_final_r4<SWZ_FIELD2>, (const void*)k_ntt_final_r4<SWZ_REV>,
                              (const void*)k_ntt_final_r4s<SWZ_FIELD>, (const void*)k_ntt_final_r4s<SWZ_FIELD2>, (const void*)k_ntt_final_r4s<SWZ_REV>})
            if (attr_err == hipSuccess && ctx->opt.ntt_lds_pad > 0) attr_err = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    ZK_HIP(attr_err);
    // pass plan
    // Up to 9 bits per pass (T = 4..8 elements per contiguous run).  Measured on MI355X: 6-bit passes with
    // 1 KiB runs are 7 % SLOWER at 2^24 than 8/9-bit passes — the kernels are bound by per-element work
    // (butterfly products, the inter-pass twiddle, canonical stores), not by the run length.
    // Round 2, isolated batches of 8 (tools/ntt_plan_bench.py): two passes of up to 11 bits beat three of 6-7 bits up to 2^21 (2^19 -10 %,
    // 2^20 -14 %, 2^21 -6 %: a polynomial is <= 64 MiB and the 32-64 byte runs of the second pass still hit the last-level cache);
    // at 2^22 11 + 11 is 1.4 % slower than 8 + 7 + 7, from 2^23 both plans are three passes.
    // register-tiled kernels (every pass on full tiles; true whenever np > 1 with the default plans).  ntt_r8: 1 = 8 elements per thread on
    // 2048-element tiles; 2 = 4 per thread, same tiles (512 threads); 3 = 4 per thread on 1024-element tiles (256 threads, 36 KiB of LDS: four
    // independent tiles per CU instead of two)
    uint32_t np = 1, sw[6] = {0, 0, 0, 0, 0, 0};
    auto plan = [&](uint32_t tlog, uint32_t sb) -> bool {
        uint32_t smax = m >= 22 ? 10 : 11;
        { int v = ctx->opt.ntt_smax; if (v >= 4 && v <= 11) smax = (uint32_t)v; }
        smax = std::min(smax, tlog);
        np = m <= 11 ? 1 : (m + smax - 1) / smax;
        for (uint32_t q = 0; q < 6; ++q) sw[q] = q < np ? m / np + (q < m % np ? 1 : 0) : 0;
        if (np > 6) return false;
        bool ok = np > 1 && twh->bf_bits >= 2;
        uint32_t lb = m;
        for (uint32_t q = 0; q < np && ok; ++q) {
            lb -= sw[q];
            uint32_t avail = q + 1 < np ? lb : sw[0];
            if (sw[q] < sb || sw[q] > tlog || avail < tlog - sw[q]) ok = false;
        }
        return ok;
    };
    // auto (4): 4 per thread everywhere — on 2048-element tiles from 2^21 (two passes of <= 11 bits), on 1024-element tiles below.  Measured in
    // the proofs (r03, A/B on one box): k = 22 -1.5 % with the 2048 tiles, k = 19 -1.0 % and k = 17 -0.6 % with the 1024 ones, against 8 per thread
    int want = ctx->opt.ntt_r8;
    if (want == 4) want = m >= 21 ? 2 : 3;
    bool r8 = false;
    uint32_t tlog = ilog2(NTT_TILE), SBh = 3;
    if (want == 3 && plan(10, 2)) { r8 = true; tlog = 10; SBh = 2; }
    else if (want == 2 && plan(tlog, 2)) { r8 = true; SBh = 2; }
    else { r8 = plan(tlog, 3) && want != 0; }
    if (np > 6) { set_error("ntt: too many passes"); return ZKHIP_EINVAL; }
    const bool r4 = r8 && SBh == 2;
    const unsigned rthreads = (1u << tlog) >> SBh;
    const size_t rlds = std::min<size_t>(((size_t)1 << tlog) * sizeof(fe) + (r4 && ctx->opt.ntt_lds_pad > 0 ? (size_t)ctx->opt.ntt_lds_pad * 1024 : 0), 160 * 1024);
    std::vector<void*> tmp_host(npolys);
    if (np > 1) {
        void* d_tmpbuf;
        ZK_TRY(ctx->get_scratch("ntt_tmp", npolys * n * 32, &d_tmpbuf));
        for (size_t i = 0; i < npolys; ++i) tmp_host[i] = (char*)d_tmpbuf + i * n * 32;
    }
    std::vector<const void*> all(3 * npolys);
    for (size_t i = 0; i < npolys; ++i) { all[i] = srcs[i]; all[npolys + i] = dsts[i]; all[2 * npolys + i] = np > 1 ? tmp_host[i] : dsts[i]; }
    if (!r8) ZK_TRY(ctx->upload(d_ptrs, all.data(), 3 * npolys * sizeof(void*)));   // the r8 kernels get their pointers by value
    NttScale sc = sc_in;
    uint32_t lo_bits = m;
    // passes 1..np-1: in place on dst, except that the first reads src and the last non-final writes tmp
    const uint32_t** cur_src = d_src;
    for (uint32_t q = 0; q + 1 < np; ++q) {
        uint32_t s = sw[q];
        lo_bits -= s;
        uint32_t logT = std::min<uint32_t>(tlog - s, lo_bits);
        NttScale scq = sc;
        if (q != 0) { scq.use_pre = 0; scq.pre_tab = nullptr; }
        scq.use_post = 0;
        scq.post_tab = nullptr;
        uint32_t** out = (q + 2 == np) ? d_tmp : d_dst;
        unsigned blocks = (unsigned)(n >> (s + logT));
        // polynomials per workgroup: as many as keep >= 512 workgroups in the launch (two per CU; measured crossover)
        uint32_t group = 1;
        while (group < npolys && (size_t)blocks * ((npolys + 2 * group - 1) / (2 * group)) >= 512) group *= 2;
        { int v = ctx->opt.ntt_group; if (v >= 1 && v <= 64) group = (uint32_t)v; }
        ProfScope ps(ctx, "ntt_strided");
        if (r8) {
            Swz z{logT + SBh, logT < 5 ? (1u << (5 - logT)) - 1 : 0u, logT};
            const void* ptab;
            ZK_TRY(pass_twiddles(ctx, twh, tw, q, s, lo_bits, &ptab));
            // host views of this pass's sources / destinations (the same choice as cur_src / out below)
            const void* const* hs = q == 0 ? all.data() : all.data() + npolys;
            const void* const* hd = (q + 2 == np) ? all.data() + 2 * npolys : all.data() + npolys;
            for (size_t p0 = 0; p0 < npolys; p0 += NTT_MAXP) {
                NttPtrs PT;
                const size_t cnt = std::min<size_t>(NTT_MAXP, npolys - p0);
                for (size_t i = 0; i < cnt; ++i) { PT.src[i] = (const uint32_t*)hs[p0 + i]; PT.dst[i] = (uint32_t*)hd[p0 + i]; }
                scq.tab_p0 = (uint32_t)p0;
                if (r4)
                    hipLaunchKernelGGL(tlog == 10 ? k_ntt_strided_r4s : k_ntt_strided_r4, dim3(blocks, (unsigned)cnt), dim3(rthreads), rlds, st, PT, m, s, lo_bits, logT,
                                       q == 0 ? n_in : (uint32_t)n, tw, scq, z, (const uint32_t*)ptab);
                else
                    hipLaunchKernelGGL(k_ntt_strided_r8, dim3(blocks, (unsigned)cnt), dim3(256), NTT_TILE * sizeof(fe), st, PT, m, s, lo_bits, logT,
                                       q == 0 ? n_in : (uint32_t)n, tw, scq, z, (const uint32_t*)ptab);
            }
        } else
        hipLaunchKernelGGL(k_ntt_strided, dim3(blocks, (unsigned)((npolys + group - 1) / group)), dim3(256), NTT_TILE * sizeof(fe), st,
                           (const uint32_t* const*)cur_src, (uint32_t* const*)out, (uint32_t)npolys, group, m, s, lo_bits, logT,
                           q == 0 ? n_in : (uint32_t)n, tw, scq);
        cur_src = (const uint32_t**)out;
    }
    {
        uint32_t s = sw[np - 1];
        uint32_t s1 = np > 1 ? sw[0] : 0;
        uint32_t logT = np > 1 ? std::min<uint32_t>(tlog - s, s1) : 0;
        NttScale scq = sc;
        if (np > 1) { scq.use_pre = 0; scq.pre_tab = nullptr; }
        NttDigits dg;
        dg.np = np;
        for (int i = 0; i < 6; ++i) dg.sw[i] = sw[i];
        unsigned blocks = (unsigned)(n >> (s + logT));
        ProfScope ps(ctx, "ntt_final");
        if (r8) {
            Swz z{tlog - 5, 31, 0};
            const int rule = logT + SBh >= 5 ? SWZ_FIELD : (SBh == 2 && logT == 0 ? SWZ_REV : SWZ_FIELD2);
            typedef void (*FinalK)(const NttPtrs, uint32_t, uint32_t, uint32_t, uint32_t, TwDev, NttScale, NttDigits, Swz);
            FinalK fk;
            if (!r4) fk = rule == SWZ_FIELD ? k_ntt_final_r8<SWZ_FIELD> : k_ntt_final_r8<SWZ_FIELD2>;
            else if (tlog == 10) fk = rule == SWZ_FIELD ? k_ntt_final_r4s<SWZ_FIELD> : rule == SWZ_FIELD2 ? k_ntt_final_r4s<SWZ_FIELD2> : k_ntt_final_r4s<SWZ_REV>;
            else fk = rule == SWZ_FIELD ? k_ntt_final_r4<SWZ_FIELD> : rule == SWZ_FIELD2 ? k_ntt_final_r4<SWZ_FIELD2> : k_ntt_final_r4<SWZ_REV>;
            const void* const* hs = all.data() + 2 * npolys;   // np > 1: the last non-final pass wrote tmp
            for (size_t p0 = 0; p0 < npolys; p0 += NTT_MAXP) {
                NttPtrs PT;
                const size_t cnt = std::min<size_t>(NTT_MAXP, npolys - p0);
                for (size_t i = 0; i < cnt; ++i) { PT.src[i] = (const uint32_t*)hs[p0 + i]; PT.dst[i] = (uint32_t*)all[npolys + p0 + i]; }
                scq.tab_p0 = (uint32_t)p0;
                hipLaunchKernelGGL(fk, dim3(blocks, (unsigned)cnt), dim3(rthreads), rlds, st, PT, m, s, logT, (uint32_t)n, tw, scq, dg, z);
            }
        } else
        hipLaunchKernelGGL(k_ntt_final, dim3(blocks, (unsigned)npolys), dim3(256), NTT_TILE * sizeof(fe), st, (const uint32_t* const*)cur_src,
                           (uint32_t* const*)d_dst, m, s, logT, np == 1 ? n_in : (uint32_t)n, tw, scq, dg);
    }
    ZK_LAUNCH_CHECK();
    return ZKHIP_OK;
}

struct zkhip_domain {
    uint32_t k, extended_k, quotient_poly_degree;
    // R' form, < 2p
    el2<Fr> omega, omega_inv, extended_omega, extended_omega_inv, g_coset, g_coset_inv, ifft_divisor, extended_ifft_divisor;
    // the same four roots in ABI form (keys of the twiddle cache, and what the caller sees)
    uint64_t omega_abi[4], omega_inv_abi[4], extended_omega_abi[4], extended_omega_inv_abi[4], g_coset_abi[4];
    void* d_t_evaluations = nullptr;   // raw R' form
    uint32_t n_t = 0;
};

static NttScale no_scale() {
    NttScale s;
    memset(&s, 0, sizeof s);
    return s;
}

extern "C" {

int zkhip_fft_batch_device(zkhip_ctx* ctx, void* const* d_polys, size_t npolys, const uint64_t omega[4], uint32_t log_n) {
    if (!ctx || !d_polys || !omega) { set_error("zkhip_fft_batch_device: null argument"); return ZKHIP_EINVAL; }
    return ntt_run(ctx, (const void* const*)d_polys, d_polys, npolys, omega, log_n, 1u << log_n, no_scale());
}

int zkhip_fft(zkhip_ctx* ctx, uint64_t* a, const uint64_t omega[4], uint32_t log_n) {
    if (!ctx || !a || !omega) { set_error("zkhip_fft: null argument"); return ZKHIP_EINVAL; }
    if (log_n > 26) { set_error("zkhip_fft: log_n = %u unsupported (max 26)", log_n); return ZKHIP_EINVAL; }
    size_t bytes = ((size_t)1 << log_n) * 32;
    void* d;
    ZK_TRY(ctx->get_scratch("fft_host", bytes, &d));
    ZK_HIP(hipMemcpyAsync(d, a, bytes, hipMemcpyHostToDevice, ctx->stream));
    void* polys[1] = {d};
    ZK_TRY(zkhip_fft_batch_device(ctx, polys, 1, omega, log_n));
    ZK_HIP(hipMemcpyAsync(a, d, bytes, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(stream_wait(ctx, ctx->stream));
    return ZKHIP_OK;
}

int zkhip_domain_new(zkhip_ctx* ctx, uint32_t j, uint32_t k, const uint64_t g_coset[4], zkhip_domain** out) {
    if (!ctx || !out) { set_error("zkhip_domain_new: null argument"); return ZKHIP_EINVAL; }
    if (j < 2 || k == 0 || k > 26) { set_error("zkhip_domain_new: j = %u, k = %u out of range", j, k); return ZKHIP_EINVAL; }
    zkhip_domain* d = new zkhip_domain();
    d->k = k;
    d->quotient_poly_degree = j - 1;
    uint32_t ek = k;
    while (((uint64_t)1 << ek) < ((uint64_t)1 << k) * d->quotient_poly_degree) ++ek;
    if (ek > 26) { delete d; set_error("zkhip_domain_new: extended_k = %u unsupported (max 26)", ek); return ZKHIP_EINVAL; }
    d->extended_k = ek;
    el2<Fr> w = from_canonical_words<Fr>(FR_ROOT_OF_UNITY);
    for (uint32_t i = ek; i < FR_S; ++i) w = sqr(w);
    d->extended_omega = w;
    for (uint32_t i = k; i < ek; ++i) w = sqr(w);
    d->omega = w;
    d->omega_inv = inv<Fr>(d->omega);
    d->extended_omega_inv = inv<Fr>(d->extended_omega);
    if (g_coset) d->g_coset = from_abi<Fr>(mem_load(g_coset)); else d->g_coset = from_canonical_words<Fr>(FR_ZETA);
    d->g_coset_inv = sqr(d->g_coset);
    d->ifft_divisor = inv<Fr>(from_u64<Fr>((uint64_t)1 << k));
    d->extended_ifft_divisor = inv<Fr>(from_u64<Fr>((uint64_t)1 << ek));
    mem_store(d->omega_abi, to_abi(d->omega));
    mem_store(d->omega_inv_abi, to_abi(d->omega_inv));
    mem_store(d->extended_omega_abi, to_abi(d->extended_omega));
    mem_store(d->extended_omega_inv_abi, to_abi(d->extended_omega_inv));
    mem_store(d->g_coset_abi, to_abi(d->g_coset));
    d->n_t = 1u << (ek - k);
    std::vector<fe32> tev(d->n_t);
    el2<Fr> cur = pow_u64<Fr>(d->g_coset, (uint64_t)1 << k), step = pow_u64<Fr>(d->extended_omega, (uint64_t)1 << k);
    for (uint32_t i = 0; i < d->n_t; ++i) {
        tev[i] = fe_pack(fe_canonical<Fr>(inv<Fr>(reduce(cur - one<Fr>())).v));
        cur = cur * step;
    }
    hipError_t e = zk::dev_malloc((void**)&d->d_t_evaluations, d->n_t * 32);
    if (e != hipSuccess) { (void)hipGetLastError(); delete d; set_error("hipMalloc t_evaluations: %s", hipGetErrorString(e)); return ZKHIP_ENOMEM; }
    ZK_HIP(hipMemcpyAsync(d->d_t_evaluations, tev.data(), d->n_t * 32, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    *out = d;
    return ZKHIP_OK;
}
void zkhip_domain_free(zkhip_ctx* ctx, zkhip_domain* d) {
    if (!d) return;
    if (ctx) (void)hipStreamSynchronize(ctx->stream);
    if (d->d_t_evaluations) (void)hipFree(d->d_t_evaluations);
    delete d;
}
uint32_t zkhip_domain_k(const zkhip_domain* d) { return d->k; }
uint32_t zkhip_domain_extended_k(const zkhip_domain* d) { return d->extended_k; }
uint32_t zkhip_domain_quotient_poly_degree(const zkhip_domain* d) { return d->quotient_poly_degree; }
void zkhip_domain_constants(const zkhip_domain* d, uint64_t omega[4], uint64_t extended_omega[4], uint64_t g_coset[4]) {
    if (omega) memcpy(omega, d->omega_abi, 32);
    if (extended_omega) memcpy(extended_omega, d->extended_omega_abi, 32);
    if (g_coset) memcpy(g_coset, d->g_coset_abi, 32);
}

int zkhip_coeff_to_lagrange_device(zkhip_ctx* ctx, const zkhip_domain* d, void* const* polys, size_t npolys) {
    if (!ctx || !d || !polys) { set_error("zkhip_coeff_to_lagrange_device: null argument"); return ZKHIP_EINVAL; }
    return ntt_run(ctx, (const void* const*)polys, polys, npolys, d->omega_abi, d->k, 1u << d->k, no_scale());
}
int zkhip_lagrange_to_coeff_device(zkhip_ctx* ctx, const zkhip_domain* d, void* const* polys, size_t npolys) {
    if (!ctx || !d || !polys) { set_error("zkhip_lagrange_to_coeff_device: null argument"); return ZKHIP_EINVAL; }
    NttScale sc = no_scale();
    sc.use_post = 1;
    sc.post[0] = sc.post[1] = sc.post[2] = d->ifft_divisor.v;
    return ntt_run(ctx, (const void* const*)polys, polys, npolys, d->omega_inv_abi, d->k, 1u << d->k, sc);
}
}  // extern "C" (reopened below)
// lagrange_to_coeff from src to dst (src untouched): the prover keeps the Lagrange form too, so this replaces a copy + in-place pass
namespace zk {
int lagrange_to_coeff_oop(zkhip_ctx* ctx, const zkhip_domain* d, const void* const* srcs, void* const* dsts, size_t npolys) {
    if (!ctx || !d || !srcs || !dsts) { set_error("lagrange_to_coeff_oop: null argument"); return ZKHIP_EINVAL; }
    NttScale sc = no_scale();
    sc.use_post = 1;
    sc.post[0] = sc.post[1] = sc.post[2] = d->ifft_divisor.v;
    return ntt_run(ctx, srcs, dsts, npolys, d->omega_inv_abi, d->k, 1u << d->k, sc);
}
// a size-2^log_n transform with per-index scaling tables (raw R' powers from power_table), either of which may be null
// (polynomial p uses table p / tab_group of a run of tables tab_stride elements apart)
int ntt_tabled(zkhip_ctx* ctx, const void* const* srcs, void* const* dsts, size_t npolys, const uint64_t omega[4], uint32_t log_n,
               const void* d_pre_tab, const void* d_post_tab, uint32_t tab_group, size_t tab_stride) {
    if (!ctx || !srcs || !dsts || !omega || !tab_group) { set_error("ntt_tabled: null argument"); return ZKHIP_EINVAL; }
    NttScale sc = no_scale();
    sc.pre_tab = (const uint32_t*)d_pre_tab;
    sc.post_tab = (const uint32_t*)d_post_tab;
    sc.tab_group = tab_group;
    sc.tab_stride = (uint32_t)tab_stride;
    return ntt_run(ctx, srcs, dsts, npolys, omega, log_n, 1u << log_n, sc);
}
// d_out[i] = base^i for i < count, in the kernels' raw R' form (32 B each), on the context's stream
int power_table(zkhip_ctx* ctx, const uint64_t base_abi[4], size_t count, void* d_out) {
    if (!ctx || !base_abi || !d_out) { set_error("power_table: null argument"); return ZKHIP_EINVAL; }
    fe32 base;   // the caller's array may be 8-byte aligned only
    memcpy(base.w, base_abi, 32);
    el2<Fr> b = from_abi<Fr>(base);
    hipLaunchKernelGGL(k_powers, dim3(div_up(div_up(count, 16), 64)), dim3(64), 0, ctx->stream, (uint32_t*)d_out, count, b.v);
    ZK_LAUNCH_CHECK();
    return ZKHIP_OK;
}
}  // namespace zk
extern "C" {
int zkhip_coeff_to_extended_device(zkhip_ctx* ctx, const zkhip_domain* d, const void* const* in, size_t n_in, void* const* out,
                                   size_t npolys) {
    if (!ctx || !d || !in || !out) { set_error("zkhip_coeff_to_extended_device: null argument"); return ZKHIP_EINVAL; }
    if (n_in > ((size_t)1 << d->extended_k)) { set_error("zkhip_coeff_to_extended_device: n_in too large"); return ZKHIP_EINVAL; }
    NttScale sc = no_scale();
    sc.use_pre = 1;  // distribute_powers_zeta(into_coset = true): [1, g, g^2] by i mod 3
    sc.pre[0] = one<Fr>().v; sc.pre[1] = d->g_coset.v; sc.pre[2] = d->g_coset_inv.v;
    return ntt_run(ctx, in, out, npolys, d->extended_omega_abi, d->extended_k, (uint32_t)n_in, sc);
}
int zkhip_extended_to_coeff_device(zkhip_ctx* ctx, const zkhip_domain* d, void* const* polys, size_t npolys) {
    if (!ctx || !d || !polys) { set_error("zkhip_extended_to_coeff_device: null argument"); return ZKHIP_EINVAL; }
    NttScale sc = no_scale();
    sc.use_post = 1;  // ifft divisor, then distribute_powers_zeta(into_coset = false): [1, g^-1, g^-2]
    sc.post[0] = d->extended_ifft_divisor.v;
    sc.post[1] = (d->extended_ifft_divisor * d->g_coset_inv).v;
    sc.post[2] = (d->extended_ifft_divisor * d->g_coset).v;
    return ntt_run(ctx, (const void* const*)polys, polys, npolys, d->extended_omega_inv_abi, d->extended_k, 1u << d->extended_k, sc);
}
int zkhip_divide_by_vanishing_device(zkhip_ctx* ctx, const zkhip_domain* d, void* d_a) {
    if (!ctx || !d || !d_a) { set_error("zkhip_divide_by_vanishing_device: null argument"); return ZKHIP_EINVAL; }
    size_t n = (size_t)1 << d->extended_k;
    hipLaunchKernelGGL(k_mul_periodic, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, (uint32_t*)d_a, n,
                       (const uint32_t*)d->d_t_evaluations, d->n_t - 1);
    ZK_LAUNCH_CHECK();
    return ZKHIP_OK;
}

// host-pointer conveniences
int zkhip_lagrange_to_coeff(zkhip_ctx* ctx, const zkhip_domain* d, uint64_t* a) {
    if (!ctx || !d || !a) { set_error("zkhip_lagrange_to_coeff: null argument"); return ZKHIP_EINVAL; }
    size_t bytes = ((size_t)1 << d->k) * 32;
    void* dev;
    ZK_TRY(ctx->get_scratch("fft_host", bytes, &dev));
    ZK_HIP(hipMemcpyAsync(dev, a, bytes, hipMemcpyHostToDevice, ctx->stream));
    void* polys[1] = {dev};
    ZK_TRY(zkhip_lagrange_to_coeff_device(ctx, d, polys, 1));
    ZK_HIP(hipMemcpyAsync(a, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(stream_wait(ctx, ctx->stream));
    return ZKHIP_OK;
}
int zkhip_coeff_to_extended(zkhip_ctx* ctx, const zkhip_domain* d, const uint64_t* coeffs, size_t n_in, uint64_t* out) {
    if (!ctx || !d || !coeffs || !out) { set_error("zkhip_coeff_to_extended: null argument"); return ZKHIP_EINVAL; }
    size_t en = (size_t)1 << d->extended_k;
    if (n_in > en) { set_error("zkhip_coeff_to_extended: n_in too large"); return ZKHIP_EINVAL; }
    void *din, *dout;
    ZK_TRY(ctx->get_scratch("fft_host_in", (n_in ? n_in : 1) * 32, &din));
    ZK_TRY(ctx->get_scratch("fft_host", en * 32, &dout));
    if (n_in) ZK_HIP(hipMemcpyAsync(din, coeffs, n_in * 32, hipMemcpyHostToDevice, ctx->stream));
    const void* ins[1] = {din};
    void* outs[1] = {dout};
    ZK_TRY(zkhip_coeff_to_extended_device(ctx, d, ins, n_in, outs, 1));
    ZK_HIP(hipMemcpyAsync(out, dout, en * 32, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(stream_wait(ctx, ctx->stream));
    return ZKHIP_OK;
}
int zkhip_extended_to_coeff(zkhip_ctx* ctx, const zkhip_domain* d, uint64_t* a) {
    if (!ctx || !d || !a) { set_error("zkhip_extended_to_coeff: null argument"); return ZKHIP_EINVAL; }
    size_t bytes = ((size_t)1 << d->extended_k) * 32;
    void* dev;
    ZK_TRY(ctx->get_scratch("fft_host", bytes, &dev));
    ZK_HIP(hipMemcpyAsync(dev, a, bytes, hipMemcpyHostToDevice, ctx->stream));
    void* polys[1] = {dev};
    ZK_TRY(zkhip_extended_to_coeff_device(ctx, d, polys, 1));
    ZK_HIP(hipMemcpyAsync(a, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(stream_wait(ctx, ctx->stream));
    return ZKHIP_OK;
}

}  // extern "C"
