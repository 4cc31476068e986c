// permute.hip — the lookup argument's permuted input / table columns on the GPU.
//
// Restates halo2_proofs plonk/lookup/prover.rs permute_expression_pair [UPSTREAM-RECALL; crate pinned at
// /root/reference/Cargo.lock:1320-1322]: over the usable rows, A' = the input column sorted by field-element order
// (canonical integer), S'[i] = A'[i] where A'[i] starts a run, and the table values not consumed that way fill the
// remaining rows (ascending value into descending row).  The result is unique, so the method is free:
//   1. both columns -> canonical integers, padded with an all-ones sentinel to n = 2^k, bitonic-sorted
//      (2048-element tiles in LDS, the large strides in global passes);
//   2. run starts, one binary search per distinct input value into the sorted table (marks the consumed instance,
//      raises ConstraintSystemFailure if absent), two exclusive scans, a compaction and a gather.
#include <algorithm>

#include "common.hpp"
using namespace zk;

#define BT_TILE 2048u   // elements per LDS tile (64 KiB), 256 threads

struct key256 { uint32_t w[8]; };   // canonical integer, w[7] most significant

__device__ __forceinline__ bool key_less(const key256& a, const key256& b) {
#pragma unroll
    for (int i = 7; i >= 0; --i) {
        if (a.w[i] != b.w[i]) return a.w[i] < b.w[i];
    }
    return false;
}
__device__ __forceinline__ bool key_eq(const key256& a, const key256& b) {
    uint32_t d = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) d |= a.w[i] ^ b.w[i];
    return d == 0;
}
__device__ __forceinline__ key256 key_load(const uint32_t* p) {
    fe32 m = mem_load(p);
    key256 k;
#pragma unroll
    for (int i = 0; i < 8; ++i) k.w[i] = m.w[i];
    return k;
}
__device__ __forceinline__ void key_store(uint32_t* p, const key256& k) {
    fe32 m;
#pragma unroll
    for (int i = 0; i < 8; ++i) m.w[i] = k.w[i];
    mem_store(p, m);
}

// ABI column -> canonical integers; rows >= usable get the sentinel 2^256 - 1 (sorts last)
__global__ void k_pe_keys(const uint32_t* col, size_t n, size_t usable, uint32_t* keys) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe32 m;
    if (i < usable) m = abi_to_canonical_words<Fr>(mem_load(col + i * 8));
    else {
#pragma unroll
        for (int j = 0; j < 8; ++j) m.w[j] = 0xffffffffu;
    }
    mem_store(keys + i * 8, m);
}

// Bitonic network: for (k = 2; k <= n; k <<= 1) for (j = k >> 1; j > 0; j >>= 1) compare-exchange(i, i ^ j), ascending iff (i & k) == 0.
// k_bitonic_tile runs, inside one 2048-element tile held in LDS, every step (k, j) with k_lo <= k <= k_hi and j < 2048
// (for k > 2048 only the j < 2048 tail of that k).
// blockIdx.y selects one of several equally sized key arrays laid out back to back (the input and the table column sort together)
__global__ void __launch_bounds__(256) k_bitonic_tile(uint32_t* keys_all, size_t n, uint32_t k_lo, uint32_t k_hi) {
    __shared__ key256 t[BT_TILE];
    uint32_t* keys = keys_all + (size_t)blockIdx.y * n * 8;
    const uint32_t tid = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * BT_TILE;
    for (uint32_t e = tid; e < BT_TILE; e += 256) t[e] = key_load(keys + (base + e) * 8);
    __syncthreads();
    for (uint32_t k = k_lo; k <= k_hi; k <<= 1) {
        for (uint32_t j = min(k >> 1, BT_TILE >> 1); j > 0; j >>= 1) {
            for (uint32_t p = tid; p < BT_TILE / 2; p += 256) {
                uint32_t i = ((p & ~(j - 1)) << 1) | (p & (j - 1));   // index with bit j clear
                uint32_t l = i | j;
                bool asc = (((base + i) & k) == 0);
                key256 a = t[i], b = t[l];
                if (key_less(b, a) == asc) { t[i] = b; t[l] = a; }
            }
            __syncthreads();
        }
    }
    for (uint32_t e = tid; e < BT_TILE; e += 256) key_store(keys + (base + e) * 8, t[e]);
}
// one global step (k, j) with j >= 2048
__global__ void k_bitonic_global(uint32_t* keys_all, size_t n, uint32_t k, uint32_t j) {
    uint32_t* keys = keys_all + (size_t)blockIdx.y * n * 8;
    size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (p >= n / 2) return;
    size_t i = ((p & ~((size_t)j - 1)) << 1) | (p & ((size_t)j - 1));
    size_t l = i | j;
    bool asc = ((i & k) == 0);
    key256 a = key_load(keys + i * 8), b = key_load(keys + l * 8);
    if (key_less(b, a) == asc) { key_store(keys + i * 8, b); key_store(keys + l * 8, a); }
}

// two global steps (k, j) and (k, j / 2) in one pass: each thread owns the four keys i + {0, j/2, j, 3j/2}
__global__ void k_bitonic_global2(uint32_t* keys_all, size_t n, uint32_t k, uint32_t j) {
    uint32_t* keys = keys_all + (size_t)blockIdx.y * n * 8;
    size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (p >= n / 4) return;
    const size_t h = j >> 1;
    size_t i = ((p & ~(h - 1)) << 2) | (p & (h - 1));   // bits j and j/2 clear
    bool asc = ((i & k) == 0);
    key256 a = key_load(keys + i * 8), b = key_load(keys + (i + h) * 8), c = key_load(keys + (i + j) * 8), d = key_load(keys + (i + j + h) * 8);
    auto cx = [&](key256& x, key256& y) { if (key_less(y, x) == asc) { key256 t = x; x = y; y = t; } };
    cx(a, c); cx(b, d);   // step j
    cx(a, b); cx(c, d);   // step j / 2
    key_store(keys + i * 8, a); key_store(keys + (i + h) * 8, b); key_store(keys + (i + j) * 8, c); key_store(keys + (i + j + h) * 8, d);
}

static int bitonic_sort(zkhip_ctx* ctx, void* d_keys, size_t n, unsigned narrays) {
    hipStream_t st = ctx->stream;
    if (n < BT_TILE || (n & (n - 1))) { set_error("bitonic_sort: n must be a power of two >= %u", BT_TILE); return ZKHIP_EINVAL; }
    unsigned tiles = (unsigned)(n / BT_TILE);
    hipLaunchKernelGGL(k_bitonic_tile, dim3(tiles, narrays), dim3(256), 0, st, (uint32_t*)d_keys, n, 2u, BT_TILE);
    for (size_t k = 2 * BT_TILE; k <= n; k <<= 1) {
        for (size_t j = k >> 1; j >= BT_TILE;) {
            if ((j >> 1) >= BT_TILE) {
                hipLaunchKernelGGL(k_bitonic_global2, dim3(div_up(n / 4, 256), narrays), dim3(256), 0, st, (uint32_t*)d_keys, n, (uint32_t)k, (uint32_t)j);
                j >>= 2;
            } else {
                hipLaunchKernelGGL(k_bitonic_global, dim3(div_up(n / 2, 256), narrays), dim3(256), 0, st, (uint32_t*)d_keys, n, (uint32_t)k, (uint32_t)j);
                j >>= 1;
            }
        }
        hipLaunchKernelGGL(k_bitonic_tile, dim3(tiles, narrays), dim3(256), 0, st, (uint32_t*)d_keys, n, (uint32_t)k, (uint32_t)k);
    }
    ZK_LAUNCH_CHECK();
    return ZKHIP_OK;
}

// first[row] (as !first -> rep flag) and the consumed table instances
__global__ void k_pe_mark(const uint32_t* A, const uint32_t* T, size_t usable, uint32_t* rep_flag, uint32_t* left_flag, uint32_t* err) {
    size_t row = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (row >= usable) return;
    key256 a = key_load(A + row * 8);
    bool first = row == 0 || !key_eq(a, key_load(A + (row - 1) * 8));
    rep_flag[row] = first ? 0u : 1u;
    if (!first) return;
    size_t lo = 0, hi = usable;   // first table entry >= a
    while (lo < hi) {
        size_t mid = (lo + hi) >> 1;
        if (key_less(key_load(T + mid * 8), a)) lo = mid + 1; else hi = mid;
    }
    if (lo >= usable || !key_eq(key_load(T + lo * 8), a)) { atomicOr(err, 1u); return; }
    left_flag[lo] = 0u;   // consumed (left_flag was initialised to 1)
}
__global__ void k_fill_u32(uint32_t* p, size_t n, uint32_t v) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
// generic exclusive scan of u32 flags: block sums, then apply (blocks of 2048)
__global__ void __launch_bounds__(256) k_scan_sums(const uint32_t* in, size_t n, uint32_t* sums) {
    __shared__ uint32_t w[4];
    size_t lo = (size_t)blockIdx.x * 2048 + threadIdx.x * 8;
    uint32_t s = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += lo + q < n ? in[lo + q] : 0u;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) s += __shfl_xor(s, d);
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = w[0] + w[1] + w[2] + w[3];
}
__global__ void __launch_bounds__(256) k_scan_apply(const uint32_t* in, size_t n, const uint32_t* sums, uint32_t nblk, uint32_t* out, uint32_t* total) {
    __shared__ uint32_t w[4], s_base;
    uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (wave == 0) {
        uint32_t b = 0, tot = 0;
        for (uint32_t j = lane; j < nblk; j += 64) { uint32_t v = sums[j]; if (j < blockIdx.x) b += v; tot += v; }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { b += __shfl_xor(b, d); tot += __shfl_xor(tot, d); }
        if (lane == 0) { s_base = b; if (blockIdx.x == 0 && total) *total = tot; }
    }
    size_t lo = (size_t)blockIdx.x * 2048 + t * 8;
    uint32_t v[8], s = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) { v[q] = lo + q < n ? in[lo + q] : 0u; s += v[q]; }
    uint32_t inc = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { uint32_t u = __shfl_up(inc, d); if ((int)lane >= d) inc += u; }
    if (lane == 63) w[wave] = inc;
    __syncthreads();
    uint32_t r = s_base + inc - s;
    for (uint32_t q = 0; q < wave; ++q) r += w[q];
#pragma unroll
    for (int q = 0; q < 8; ++q) { if (lo + q < n) out[lo + q] = r; r += v[q]; }
}
static int scan_u32(zkhip_ctx* ctx, const void* d_in, size_t n, void* d_out, void* d_total, const char* tag) {
    unsigned nblk = div_up(n, 2048);
    void* d_sums;
    ZK_TRY(ctx->get_scratch(tag, nblk * 4, &d_sums));
    hipLaunchKernelGGL(k_scan_sums, dim3(nblk), dim3(256), 0, ctx->stream, (const uint32_t*)d_in, n, (uint32_t*)d_sums);
    hipLaunchKernelGGL(k_scan_apply, dim3(nblk), dim3(256), 0, ctx->stream, (const uint32_t*)d_in, n, (const uint32_t*)d_sums, nblk,
                       (uint32_t*)d_out, (uint32_t*)d_total);
    return ZKHIP_OK;
}
__global__ void k_pe_compact(const uint32_t* T, const uint32_t* left_flag, const uint32_t* left_rank, size_t usable, uint32_t* leftover) {
    size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t >= usable || !left_flag[t]) return;
    key_store(leftover + (size_t)left_rank[t] * 8, key_load(T + t * 8));
}
// canonical integer words -> ABI form
__device__ __forceinline__ void store_from_canonical(uint32_t* p, const key256& k) {
    mem_store(p, to_abi(from_canonical_words<Fr>(k.w)));
}
__global__ void k_pe_finish(const uint32_t* A, const uint32_t* leftover, const uint32_t* rep_flag, const uint32_t* rep_rank,
                            const uint32_t* totals /* [0] = #repeated, [1] = #leftover */, size_t n, size_t usable,
                            const uint32_t* blind_in, const uint32_t* blind_tab, uint32_t* perm_in, uint32_t* perm_tab, uint32_t* err) {
    size_t row = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (row >= n) return;
    if (row >= usable) {
        mem_store(perm_in + row * 8, mem_load(blind_in + (row - usable) * 8));
        mem_store(perm_tab + row * 8, mem_load(blind_tab + (row - usable) * 8));
        return;
    }
    if (row == 0 && totals[0] != totals[1]) atomicOr(err, 2u);
    key256 a = key_load(A + row * 8);
    store_from_canonical(perm_in + row * 8, a);
    if (!rep_flag[row]) { store_from_canonical(perm_tab + row * 8, a); return; }
    uint32_t R = totals[0];
    uint32_t idx = R - 1 - rep_rank[row];   // ascending leftover value -> descending repeated row
    if (idx >= totals[1]) return;           // inconsistent counts: flagged above
    store_from_canonical(perm_tab + row * 8, key_load(leftover + (size_t)idx * 8));
}

// ---- fast path for a table whose sorted keys are already there (a fixed range table): every input value must occur in the table,
// so its position in the SORTED TABLE is a small integer sort key — one binary search per row, then a counting sort, instead of a
// bitonic network over 256-bit keys.
// rank[row] = first sorted-table index holding the row's value (err |= 1 if there is none); hist[rank]++
__global__ void k_pe_rank(const uint32_t* input, const uint32_t* T, size_t usable, uint32_t* rank, uint32_t* hist, uint32_t* err) {
    size_t row = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (row >= usable) return;
    fe32 m = abi_to_canonical_words<Fr>(mem_load(input + row * 8));
    key256 a;
#pragma unroll
    for (int i = 0; i < 8; ++i) a.w[i] = m.w[i];
    size_t lo = 0, hi = usable;   // first table entry >= a
    while (lo < hi) {
        size_t mid = (lo + hi) >> 1;
        if (key_less(key_load(T + mid * 8), a)) lo = mid + 1; else hi = mid;
    }
    if (lo >= usable || !key_eq(key_load(T + lo * 8), a)) { atomicOr(err, 1u); rank[row] = 0xffffffffu; return; }
    rank[row] = (uint32_t)lo;
    atomicAdd(&hist[lo], 1u);
}
// index t plays two roles: as a ROW it scatters its key to its place in the sorted input (start[rank] + arrival order; equal keys
// are interchangeable) and flags the place as a repeat unless it is the run's first; as a TABLE SLOT it is left over iff no row
// took its value
__global__ void k_pe_place(const uint32_t* T, const uint32_t* rank, const uint32_t* hist, const uint32_t* start, uint32_t* cursor,
                           size_t usable, uint32_t* A_sorted, uint32_t* rep_flag, uint32_t* left_flag) {
    size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t >= usable) return;
    left_flag[t] = hist[t] ? 0u : 1u;
    uint32_t r = rank[t];
    if (r == 0xffffffffu) return;   // not in the table: the failure flag is already set, the outputs are void
    uint32_t s0 = start[r];
    uint32_t pos = s0 + atomicAdd(&cursor[r], 1u);
    key_store(A_sorted + (size_t)pos * 8, key_load(T + (size_t)r * 8));
    rep_flag[pos] = pos != s0 ? 1u : 0u;
}

namespace zk {
// Asynchronous form for the library's own schedule: the failure flag (non-zero = ConstraintSystemFailure) is OR-ed into *d_err_flag,
// which the caller zeroes beforehand and reads at its next synchronisation point.
int permute_expression_pair_async(zkhip_ctx* ctx, uint32_t k, uint32_t blinding_factors, const void* d_input, const void* d_table,
                                  const void* d_blind_in, const void* d_blind_tab, void* d_perm_in, void* d_perm_tab, uint32_t* d_err_flag,
                                  const void* d_sorted_table_keys);
int permute_sorted_table_keys(zkhip_ctx* ctx, uint64_t key_id, uint32_t slot, uint32_t k, uint32_t blinding_factors, const void* d_table,
                              const void** d_keys);
}
extern "C" int zkhip_permute_expression_pair_device(zkhip_ctx* ctx, uint32_t k, uint32_t blinding_factors, const void* d_input,
                                                    const void* d_table, const void* d_blind_in, const void* d_blind_tab,
                                                    void* d_perm_in, void* d_perm_tab) {
    if (!ctx) { set_error("zkhip_permute_expression_pair_device: null argument"); return ZKHIP_EINVAL; }
    void* d_err;
    ZK_TRY(ctx->get_scratch("pe_err", 16, &d_err));
    ZK_HIP(hipMemsetAsync(d_err, 0, 16, ctx->stream));
    ZK_TRY(zk::permute_expression_pair_async(ctx, k, blinding_factors, d_input, d_table, d_blind_in, d_blind_tab, d_perm_in, d_perm_tab,
                                             (uint32_t*)d_err, nullptr));
    uint32_t* h_err = (uint32_t*)ctx->h_pinned;
    ZK_HIP(hipMemcpyAsync(h_err, d_err, 4, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(stream_wait(ctx, ctx->stream));
    if (*h_err) { set_error("permute_expression_pair: an input value is not in the table (ConstraintSystemFailure)"); return ZKHIP_ECONSTRAINT; }
    return ZKHIP_OK;
}
// The sorted canonical keys of a table column that never changes (a fixed column used as a single-expression lookup table: the
// range table of halo2-lib), computed once per (proving key id, lookup, k, blinding_factors) and kept by the context.
int zk::permute_sorted_table_keys(zkhip_ctx* ctx, uint64_t key_id, uint32_t slot, uint32_t k, uint32_t blinding_factors, const void* d_table,
                                  const void** d_keys) {
    char name[96];
    snprintf(name, sizeof name, "pe_table_keys:%llx:%u:%u:%u", (unsigned long long)key_id, slot, k, blinding_factors);
    auto it = ctx->persistent.find(name);
    if (it != ctx->persistent.end()) { *d_keys = it->second; return ZKHIP_OK; }
    size_t n = (size_t)1 << k, np = std::max(n, (size_t)BT_TILE), usable = n - (blinding_factors + 1);
    void* d = nullptr;
    hipError_t e = zk::dev_malloc((void**)&d, np * 32);
    if (e != hipSuccess) { (void)hipGetLastError(); set_error("hipMalloc(%zu) for the sorted table keys failed: %s", np * 32, hipGetErrorString(e)); return ZKHIP_ENOMEM; }
    hipLaunchKernelGGL(k_pe_keys, dim3(div_up(np, 256)), dim3(256), 0, ctx->stream, (const uint32_t*)d_table, np, usable, (uint32_t*)d);
    int rc = bitonic_sort(ctx, d, np, 1);
    if (rc != ZKHIP_OK) { (void)hipFree(d); return rc; }
    ctx->persistent[name] = d;
    *d_keys = d;
    return ZKHIP_OK;
}

int zk::permute_expression_pair_async(zkhip_ctx* ctx, uint32_t k, uint32_t blinding_factors, const void* d_input, const void* d_table,
                                      const void* d_blind_in, const void* d_blind_tab, void* d_perm_in, void* d_perm_tab, uint32_t* err,
                                      const void* d_sorted_table_keys) {
    if (!ctx || !d_input || !d_table || !d_blind_in || !d_blind_tab || !d_perm_in || !d_perm_tab || !err) { set_error("zkhip_permute_expression_pair_device: null argument"); return ZKHIP_EINVAL; }
    if (k < 1 || k > 26) { set_error("zkhip_permute_expression_pair_device: k = %u unsupported (1..26)", k); return ZKHIP_EINVAL; }
    size_t n = (size_t)1 << k;
    size_t np = std::max(n, (size_t)BT_TILE);   // sort size: short columns are padded with sentinels to one tile
    if ((size_t)blinding_factors + 1 >= n) { set_error("zkhip_permute_expression_pair_device: too many blinding factors"); return ZKHIP_EINVAL; }
    size_t usable = n - (blinding_factors + 1);
    hipStream_t st = ctx->stream;
    void *dA, *dT, *d_flags, *d_ranks, *d_left, *d_misc;
    ZK_TRY(ctx->get_scratch("pe_keys", 2 * np * 32, &dA));   // input keys, then table keys: sorted by the same launches
    dT = (char*)dA + np * 32;
    ZK_TRY(ctx->get_scratch("pe_flags", 2 * n * 4, &d_flags));
    ZK_TRY(ctx->get_scratch("pe_ranks", 2 * n * 4, &d_ranks));
    ZK_TRY(ctx->get_scratch("pe_left", n * 32, &d_left));
    ZK_TRY(ctx->get_scratch("pe_misc", 16, &d_misc));   // totals[2] (+ one unused total of the rank path)
    uint32_t* rep_flag = (uint32_t*)d_flags; uint32_t* left_flag = rep_flag + n;
    uint32_t* rep_rank = (uint32_t*)d_ranks; uint32_t* left_rank = rep_rank + n;
    uint32_t* totals = (uint32_t*)d_misc;
    ProfScope ps(ctx, "lookup_permute");
    ZK_HIP(hipMemsetAsync(d_misc, 0, 16, st));
    unsigned g = div_up(n, 256);
    const bool rank_sort = ctx->opt.permute_rank_sort != 0;
    if (d_sorted_table_keys && rank_sort) {
        // a fixed table sorted once at first use: rank every row in it and counting-sort the ranks (see k_pe_rank)
        dT = const_cast<void*>(d_sorted_table_keys);
        void* d_cnt;
        ZK_TRY(ctx->get_scratch("pe_rank", 4 * n * 4, &d_cnt));   // rank, hist, cursor, start
        uint32_t* rank = (uint32_t*)d_cnt; uint32_t* hist = rank + n; uint32_t* cursor = hist + n; uint32_t* start = cursor + n;
        ZK_HIP(hipMemsetAsync(hist, 0, 2 * n * 4, st));
        ZK_HIP(hipMemsetAsync(rep_flag, 0, n * 4, st));
        hipLaunchKernelGGL(k_pe_rank, dim3(div_up(usable, 256)), dim3(256), 0, st, (const uint32_t*)d_input, (const uint32_t*)dT, usable, rank, hist, err);
        ZK_TRY(scan_u32(ctx, hist, usable, start, totals + 2, "pe_sums_h"));
        hipLaunchKernelGGL(k_pe_place, dim3(div_up(usable, 256)), dim3(256), 0, st, (const uint32_t*)dT, rank, hist, start, cursor, usable,
                           (uint32_t*)dA, rep_flag, left_flag);
    } else {
    hipLaunchKernelGGL(k_pe_keys, dim3(div_up(np, 256)), dim3(256), 0, st, (const uint32_t*)d_input, np, usable, (uint32_t*)dA);
    if (d_sorted_table_keys) {   // a fixed table sorted once at first use: only the input column is sorted per proof
        dT = const_cast<void*>(d_sorted_table_keys);
        ZK_TRY(bitonic_sort(ctx, dA, np, 1));
    } else {
        hipLaunchKernelGGL(k_pe_keys, dim3(div_up(np, 256)), dim3(256), 0, st, (const uint32_t*)d_table, np, usable, (uint32_t*)dT);
        ZK_TRY(bitonic_sort(ctx, dA, np, 2));
    }
    hipLaunchKernelGGL(k_fill_u32, dim3(g), dim3(256), 0, st, left_flag, n, 1u);
    ZK_HIP(hipMemsetAsync(rep_flag, 0, n * 4, st));
    hipLaunchKernelGGL(k_pe_mark, dim3(div_up(usable, 256)), dim3(256), 0, st, (const uint32_t*)dA, (const uint32_t*)dT, usable, rep_flag, left_flag, err);
    }
    ZK_TRY(scan_u32(ctx, rep_flag, usable, rep_rank, totals, "pe_sums_a"));
    ZK_TRY(scan_u32(ctx, left_flag, usable, left_rank, totals + 1, "pe_sums_b"));
    hipLaunchKernelGGL(k_pe_compact, dim3(div_up(usable, 256)), dim3(256), 0, st, (const uint32_t*)dT, left_flag, left_rank, usable, (uint32_t*)d_left);
    hipLaunchKernelGGL(k_pe_finish, dim3(g), dim3(256), 0, st, (const uint32_t*)dA, (const uint32_t*)d_left, rep_flag, rep_rank, totals, n, usable,
                       (const uint32_t*)d_blind_in, (const uint32_t*)d_blind_tab, (uint32_t*)d_perm_in, (uint32_t*)d_perm_tab, err);
    ZK_LAUNCH_CHECK();
    return ZKHIP_OK;
}
