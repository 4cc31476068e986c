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
//   * the multi-open divides once per DISTINCT opening point, not once per (rotation set, point): the quotient is linear in the
//     numerator, so the sets' partial fractions are regrouped by root (zk::shplonk_open).
#include <algorithm>
#include <vector>

#include "common.hpp"
#include "hostfield.hpp"
using namespace zk;

#define SP_PER 16
#define SP_PER_LOG 4
#define SP_TILE_LOG 12
#define SP_BLOCK 256
#define SP_TILE (SP_PER * SP_BLOCK)
#define LC_MAX 64   // polynomials per linear-combination launch

// ------------------------------------------------------------------ linear combination
// Pointers, scalars and the low-degree correction travel as kernel arguments (2.9 KB): no staging copy, no host sync.
#define LC_LOW_MAX 8
struct LcArgs {
    const uint32_t* p[LC_MAX];
    fe32 c[LC_MAX];        // R' form, canonical
    fe32 low[LC_LOW_MAX];  // ABI scale; a correction of more than LC_LOW_MAX coefficients (a rotation set of more than 8 points:
                           // the zkevm SHA-256 bit circuit queries its bit columns at many rotations) comes from low_dev instead
    const uint32_t* low_dev;
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
    // four terms (dot4: 414 multiplier instructions instead of 684) or two (muladd2) share one Montgomery reduction
    for (; j + 8 <= npolys; j += 8) {   // eight loads in flight
        auto t = dot4<Fr>(load_raw<Fr>(A.p[j] + i * 8), el1<Fr>(sc[j]), load_raw<Fr>(A.p[j + 1] + i * 8), el1<Fr>(sc[j + 1]),
                          load_raw<Fr>(A.p[j + 2] + i * 8), el1<Fr>(sc[j + 2]), load_raw<Fr>(A.p[j + 3] + i * 8), el1<Fr>(sc[j + 3])) +
                 dot4<Fr>(load_raw<Fr>(A.p[j + 4] + i * 8), el1<Fr>(sc[j + 4]), load_raw<Fr>(A.p[j + 5] + i * 8), el1<Fr>(sc[j + 5]),
                          load_raw<Fr>(A.p[j + 6] + i * 8), el1<Fr>(sc[j + 6]), load_raw<Fr>(A.p[j + 7] + i * 8), el1<Fr>(sc[j + 7]));
        acc = canonical(acc + t);
    }
    for (; j + 2 <= npolys; j += 2)
        acc = canonical(acc + muladd2(load_raw<Fr>(A.p[j] + i * 8), el1<Fr>(sc[j]), load_raw<Fr>(A.p[j + 1] + i * 8), el1<Fr>(sc[j + 1])));
    for (; j < npolys; ++j) acc = canonical(acc + load_raw<Fr>(A.p[j] + i * 8) * el1<Fr>(sc[j]));
    if (i < nlow) acc = canonical(acc - el1<Fr>(fe_split<0>(A.low_dev ? mem_load(A.low_dev + i * 8) : A.low[i])));
    store_raw<Fr>(out + i * 8, acc);
}

// ------------------------------------------------------------------ division by X - r
#define KD_MAX 16   // polynomials per launch
struct KdEntry {
    const uint32_t* src;
    uint32_t* dst;   // may equal src
    fe32 r;          // root, R' form canonical
    fe32 r_per;      // r^SP_PER and r^SP_TILE, same form: filled in by divide_round on the host (16 host products) — on the device they
    fe32 r_tile;     // were 4 / 12 dependent squarings at the head of every workgroup: pure latency at 2^17, where a pass is ~40 products deep
};
struct KdArgs { KdEntry e[KD_MAX]; };

// suffix scan over the 256 per-thread values in LDS: S_t = sum_{q >= t} m^(q - t) A_q, m = the slope of one thread's span (the same
// in every thread).  Level l needs m^(2^l): the LAST thread, which has no partner in any level, squares the next level's slope while the
// others do their product — one product per thread and level, none of them ahead of the scan (until round 4 thread 0 squared all eight
// before the first level: 8 dependent products with 255 threads waiting).  The first barrier below orders the callers' sc[] writes too.
__device__ __forceinline__ void suffix_scan_256(fe* sc, fe* mp, uint32_t t, const el2<Fr>& m) {
    if (t == SP_BLOCK - 1) mp[0] = m.v;
    __syncthreads();
    int l = 0;
    for (uint32_t d = 1; d < SP_BLOCK; d <<= 1, ++l) {
        bool on = t + d < SP_BLOCK;
        fe mine = sc[t];
        fe other = on ? sc[t + d] : fe_zero();
        fe slope = mp[l];
        __syncthreads();
        if (on) sc[t] = canonical(el1<Fr>(mine) + el1<Fr>(other) * el2<Fr>(slope)).v;
        else if (t == SP_BLOCK - 1 && l < 7) mp[l + 1] = sqr(el2<Fr>(slope)).v;
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
    __shared__ fe mp[8];
    suffix_scan_256(sc, mp, t, el2<Fr>(fe_split<0>(K.e[e].r_per)));
    if (t == 0) mem_store(tot_all + ((size_t)e * nblk + blockIdx.x) * 8, fe_pack(sc[0]));
}
// pass 2 (one block per entry): carry[entry][blk] = s at the first index of tile blk + 1 (exclusive suffix scan, slope r^2048)
// If total_all is given, entry e's Horner total over ALL tiles (the value of the suffix recurrence at the array's first index with zero
// carry-in: sum_j a_j r^j) is stored there — what a rank of a row-sharded division hands to the ranks below it.
__global__ void __launch_bounds__(SP_BLOCK) k_kd_carries(const KdArgs K, uint32_t nblk, const uint32_t* tot_all, uint32_t* carry_all, uint32_t* total_all) {
    __shared__ fe sc[SP_BLOCK];
    const uint32_t t = threadIdx.x, e = blockIdx.x;
    const uint32_t* tot = tot_all + (size_t)e * nblk * 8;
    uint32_t* carry = carry_all + (size_t)e * nblk * 8;
    const el2<Fr> M(fe_split<0>(K.e[e].r_tile));   // r^SP_TILE
    const uint32_t c = (nblk + SP_BLOCK - 1) / SP_BLOCK;
    const uint32_t lo = min(nblk, t * c), hi = min(nblk, lo + c);
    el<Fr, 4 * U> s = zero<Fr>();
    for (uint32_t q = hi; q > lo; --q) s = s * M + load_raw<Fr>(tot + (size_t)(q - 1) * 8);
    sc[t] = canonical(s).v;
    __shared__ fe mp[8];
    suffix_scan_256(sc, mp, t, pow_u64<Fr>(M, c));
    if (total_all && t == 0) store_raw<Fr>(total_all + (size_t)e * 8, el1<Fr>(sc[0]));
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
#pragma clang loop unroll(full)
    for (int j = SP_PER - 1; j >= 0; --j) {
        v[j] = zero<Fr>();
        if (lo + j < n) v[j] = load_raw<Fr>(a + (lo + j) * 8);
        s = s * r + v[j];
    }
    const el2<Fr> m(fe_split<0>(K.e[e].r_per));   // r^SP_PER: the scan's slope and the weight of the tile's carry
    const el1<Fr> cb = load_raw<Fr>(carry_all + ((size_t)e * nblk + blockIdx.x) * 8);
    el1<Fr> agg = canonical(s);
    if (t == SP_BLOCK - 1) agg = canonical(agg + cb * m);   // the tile's carry enters above its last thread
    sc[t] = agg.v;
    __shared__ fe mp[8];
    suffix_scan_256(sc, mp, t, m);
    s = (t + 1 < SP_BLOCK) ? el1<Fr>(sc[t + 1]) : cb;
#pragma clang loop unroll(full)
    for (int j = SP_PER - 1; j >= 0; --j) {
        if (lo + j < n) store_raw<Fr>(dst + (lo + j) * 8, s);
        s = s * r + v[j];
    }
}

// Row-sharded division: the local array is rows [lo, lo + m) of the polynomial and the recurrence continues above it: with the carry-in
// c = s[lo + m] the true quotient is q[j] = q0[j] + c r^(m - 1 - j), q0 the zero-carry result of the three passes above.  Each thread fixes
// 16 consecutive coefficients: one power by square-and-multiply, then a running product downwards.
struct KdFix { uint32_t* dst; fe32 r; fe32 c; };   // r, c in R' form canonical
struct KdFixArgs { KdFix e[KD_MAX]; };
__global__ void __launch_bounds__(256) k_kd_fix(const KdFixArgs F, size_t m) {
    const uint32_t e = blockIdx.y;
    const size_t j0 = (blockIdx.x * (size_t)256 + threadIdx.x) * 16;
    if (j0 >= m) return;
    const size_t j1 = j0 + 16 < m ? j0 + 16 : m;
    const el1<Fr> r(fe_split<0>(F.e[e].r)), c(fe_split<0>(F.e[e].c));
    el2<Fr> w = c * pow_u64<Fr>(el2<Fr>(r), (uint64_t)(m - j1));      // c r^(m - 1 - (j1 - 1))
    uint32_t* dst = F.e[e].dst;
    for (size_t j = j1; j-- > j0;) {
        store_raw<Fr>(dst + j * 8, load_raw<Fr>(dst + j * 8) + w);
        w = w * r;
    }
}

static fe32 abi_to_raw(const uint64_t* p) { return fe_pack(fe_canonical<Fr>(from_abi<Fr>(mem_load(p)).v)); }

// dst_j = src_j / (X - root_j) for `count` polynomials, KD_MAX per launch triple.
// sharded = true (a context with a communicator, the arrays are this rank's rows [RK m, (RK + 1) m) of n = N m coefficients): the suffix
// recurrence runs across the ranks — every rank publishes its range totals (32 B per division, one all-gather), derives its carry-in
// from the ranks above it on the host (Horner in r^m) and adds the geometric correction (k_kd_fix).
static int divide_round(zkhip_ctx* ctx, size_t n, const std::vector<KdEntry>& ents, bool sharded = false) {
    hipStream_t st = ctx->stream;
    uint32_t nblk = div_up(n, SP_TILE);
    void *d_tot, *d_carry, *d_total = nullptr;
    ZK_TRY(ctx->get_scratch("kd_tot", (size_t)KD_MAX * nblk * 32, &d_tot));
    ZK_TRY(ctx->get_scratch("kd_carry", (size_t)KD_MAX * nblk * 32, &d_carry));
    const size_t NR = (size_t)ctx->comm.nranks, RK = (size_t)ctx->comm.rank;
    if (sharded) ZK_TRY(ctx->get_scratch("kd_total", (NR + 1) * KD_MAX * 32, &d_total));
    for (size_t done = 0; done < ents.size(); done += KD_MAX) {
        uint32_t cnt = (uint32_t)std::min<size_t>(KD_MAX, ents.size() - done);
        KdArgs K;
        memset(&K, 0, sizeof K);
        for (uint32_t j = 0; j < cnt; ++j) {
            K.e[j] = ents[done + j];
            HF rp = hf_from_fe32(to_abi(el1<Fr>(fe_split<0>(K.e[j].r))));
            for (int q = 0; q < SP_PER_LOG; ++q) rp = hmul(rp, rp);
            K.e[j].r_per = hf_raw(rp);
            for (int q = SP_PER_LOG; q < SP_TILE_LOG; ++q) rp = hmul(rp, rp);
            K.e[j].r_tile = hf_raw(rp);
        }
        uint32_t* mine = sharded ? (uint32_t*)((char*)d_total + RK * KD_MAX * 32) : nullptr;
        hipLaunchKernelGGL(k_kd_totals, dim3(nblk, cnt), dim3(SP_BLOCK), 0, st, K, n, nblk, (uint32_t*)d_tot);
        hipLaunchKernelGGL(k_kd_carries, dim3(cnt), dim3(SP_BLOCK), 0, st, K, nblk, (const uint32_t*)d_tot, (uint32_t*)d_carry, mine);
        hipLaunchKernelGGL(k_kd_apply, dim3(nblk, cnt), dim3(SP_BLOCK), 0, st, K, n, nblk, (const uint32_t*)d_carry);
        ZK_LAUNCH_CHECK();
        if (sharded) {
            ZK_TRY(zk::comm_allgather(ctx, mine, d_total, KD_MAX * 32));
            std::vector<uint32_t> tot(NR * KD_MAX * 8);
            ZK_TRY(zkhip_memcpy_d2h(ctx, tot.data(), d_total, NR * KD_MAX * 32));
            if (RK + 1 < NR) {
                KdFixArgs F;
                memset(&F, 0, sizeof F);
                for (uint32_t j = 0; j < cnt; ++j) {
                    // the totals are in the coefficients' scale (the ABI form, which is HF's form); the root travels in the R' form:
                    // c = T_(R+1) + r^m (T_(R+2) + r^m (...)) with r^m in the ABI form
                    fe32 rr = K.e[j].r;                          // r in R' form, canonical
                    HF r_abi = hf_from_fe32(to_abi(el1<Fr>(fe_split<0>(rr))));
                    const HF rm = hpow(r_abi, (uint64_t)n);
                    HF acc = hzero();
                    for (size_t q = NR; q-- > RK + 1;) {
                        fe32 t;
                        memcpy(t.w, tot.data() + (q * KD_MAX + j) * 8, 32);
                        acc = hadd(hmul(acc, rm), hf_from_fe32(t));
                    }
                    F.e[j].dst = K.e[j].dst;
                    F.e[j].r = rr;
                    F.e[j].c = hf_words(acc);
                }
                hipLaunchKernelGGL(k_kd_fix, dim3(div_up(div_up(n, 16), 256), cnt), dim3(256), 0, st, F, n);
                ZK_LAUNCH_CHECK();
            }
        }
    }
    ZK_LAUNCH_CHECK();
    return ZKHIP_OK;
}

// out = sum_j c_j polys[j] - low with the scalars already in R' form (raw) and low in ABI scale
static int lincomb_raw(zkhip_ctx* ctx, size_t n, const void* const* d_polys, size_t npolys, const fe32* coeffs_raw, const fe32* low_abi,
                       size_t nlow, void* d_out) {
    void* d_low = nullptr;
    if (nlow > LC_LOW_MAX) {   // stream-ordered upload through the pinned ring: the previous launch that read this scratch is ahead of it
        ZK_TRY(ctx->get_scratch("lc_low", nlow * 32, &d_low));
        ZK_TRY(ctx->upload(d_low, low_abi, nlow * 32));
    }
    ProfScope ps(ctx, "linear_combination");
    size_t done = 0;
    do {
        uint32_t cnt = (uint32_t)std::min<size_t>(LC_MAX, npolys - done);
        bool last = done + cnt == npolys;
        LcArgs A;
        memset(&A, 0, sizeof A);
        for (uint32_t j = 0; j < cnt; ++j) {
            A.p[j] = (const uint32_t*)d_polys[done + j];
            A.c[j] = coeffs_raw[done + j];
        }
        if (last && !d_low) for (size_t j = 0; j < nlow; ++j) A.low[j] = low_abi[j];
        if (last) A.low_dev = (const uint32_t*)d_low;
        hipLaunchKernelGGL(k_lincomb, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, A, cnt, n, last ? (uint32_t)nlow : 0u, done ? 1 : 0,
                           (uint32_t*)d_out);
        done += cnt;
    } while (done < npolys);
    ZK_LAUNCH_CHECK();
    return ZKHIP_OK;
}

// ------------------------------------------------------------------ ProverSHPLONK::create_proof, host side
namespace {
struct RotSet {
    std::vector<uint32_t> points;                 // indices into the unique point list, ascending by field value
    std::vector<uint32_t> commits;                // polynomial indices, query order
    std::vector<std::vector<HF>> evals;           // [commit][point]
    std::vector<HF> low;                          // [degree]: R_i(X) = sum_j y^j R_ij(X), the set's combined low-degree equivalent
    std::vector<HF> inv_den;                      // 1 / prod_{s != r} (r - s) per point
};
}  // namespace

// rows_only: the caller's polynomials exist as THIS RANK'S ROW RANGE only (zkhip_create_proof_ex with sharded pieces): the row-sharded
// form below is then not a choice but a precondition, and the call fails instead of reading rows nobody filled.
int zk::shplonk_open(zkhip_ctx* ctx, const zkhip_srs* srs, size_t n, const void* const* d_polys, size_t npolys, const uint32_t* query_poly,
                     const uint64_t* query_points, const uint64_t* query_evals, size_t nq, const zk_transcript* tr, uint64_t h1_xy[8],
                     uint64_t h2_xy[8], bool rows_only) {
    if (!ctx || !srs || !d_polys || !query_poly || !query_points || !query_evals || !tr || !tr->write_point || !tr->squeeze_challenge || !h1_xy || !h2_xy) {
        set_error("zkhip_shplonk_open: null argument");
        return ZKHIP_EINVAL;
    }
    if (nq == 0 || n == 0 || n > zkhip_srs_len(srs)) { set_error("zkhip_shplonk_open: bad sizes (n = %zu, queries = %zu)", n, nq); return ZKHIP_EINVAL; }
    // Row-sharded (a communicator, MSMs by point range): every polynomial arithmetic below is pointwise or a suffix recurrence, and a
    // point-range commitment reads only this rank's rows — so each rank works on rows [lo, lo + nl) of every polynomial: linear
    // combinations on the range, divisions with the carries exchanged as 32-byte range totals (divide_round), the low-degree correction on
    // rank 0 only.  The scratch polynomials keep their full size; only the rank's range of them is ever written or read.
    size_t lo = 0, nl = n;
    bool sharded = false;
    {
        size_t s_first, s_count, s_total;
        zkhip_srs_range(srs, &s_first, &s_count, &s_total);
        const size_t NR = (size_t)ctx->comm.nranks, RK = (size_t)ctx->comm.rank;
        if (NR > 1 && ctx->comm.row_sharded(ctx->opt) && !ctx->comm.shard_columns && n % NR == 0 && s_total == n && s_count == n / NR && s_first == RK * (n / NR) &&
            n / NR >= 64) {
            sharded = true;
            ctx->n_shplonk_sharded += 1;
            nl = n / NR;
            lo = RK * nl;
        }
        if (rows_only && !sharded) {
            set_error("zkhip_shplonk_open: the polynomials exist as row ranges only, but this context / SRS handle does not select the row-sharded multi-open "
                      "(ranks %zu, row_sharded %d, shard_columns %d, SRS range [%zu, +%zu) of %zu, n %zu)", NR, (int)ctx->comm.row_sharded(ctx->opt),
                      ctx->comm.shard_columns, s_first, s_count, s_total, n);
            return ZKHIP_EINVAL;
        }
    }
    const size_t lo_b = lo * 32;
    // ---- construct_intermediate_sets: unique points (by value), the point set of every commitment, sets in first-appearance order
    std::vector<HF> upts;
    std::vector<Words> uwords;
    std::vector<uint32_t> q_pt(nq);
    for (size_t i = 0; i < nq; ++i) {
        if (query_poly[i] >= npolys) { set_error("zkhip_shplonk_open: query %zu names polynomial %u of %zu", i, query_poly[i], npolys); return ZKHIP_EINVAL; }
        HF pnt = hf_from_abi(query_points + 4 * i);
        Words w = canon_words(pnt);
        uint32_t j = 0;
        for (; j < upts.size(); ++j) if (memcmp(uwords[j].w, w.w, 32) == 0) break;
        if (j == upts.size()) { upts.push_back(pnt); uwords.push_back(w); }
        q_pt[i] = j;
    }
    const uint32_t np = (uint32_t)upts.size();
    std::vector<uint32_t> order(np);            // point indices ascending by value = the super point set
    for (uint32_t i = 0; i < np; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return words_less(uwords[a], uwords[b]); });
    std::vector<uint32_t> rank(np);
    for (uint32_t i = 0; i < np; ++i) rank[order[i]] = i;
    struct Commit { uint32_t poly; std::vector<std::pair<uint32_t, HF>> q; };   // (point, eval), points distinct
    std::vector<Commit> commits;
    for (size_t i = 0; i < nq; ++i) {
        size_t c = 0;
        for (; c < commits.size(); ++c) if (commits[c].poly == query_poly[i]) break;
        if (c == commits.size()) commits.push_back(Commit{query_poly[i], {}});
        bool dup = false;
        for (auto& pe : commits[c].q) dup |= pe.first == q_pt[i];
        if (!dup) commits[c].q.push_back({q_pt[i], hf_from_abi(query_evals + 4 * i)});
    }
    std::vector<RotSet> sets;
    for (auto& cm : commits) {
        std::sort(cm.q.begin(), cm.q.end(), [&](const std::pair<uint32_t, HF>& a, const std::pair<uint32_t, HF>& b) { return rank[a.first] < rank[b.first]; });
        std::vector<uint32_t> pts;
        std::vector<HF> ev;
        for (auto& pe : cm.q) { pts.push_back(pe.first); ev.push_back(pe.second); }
        size_t s_ = 0;
        for (; s_ < sets.size(); ++s_) if (sets[s_].points == pts) break;
        if (s_ == sets.size()) { sets.push_back(RotSet()); sets.back().points = pts; }
        sets[s_].commits.push_back(cm.poly);
        sets[s_].evals.push_back(ev);
    }
    const size_t nsets = sets.size();
    // ---- y, v
    uint64_t ch[4];
    tr->squeeze_challenge(tr->user, ch);
    const HF y = hf_from_abi(ch);
    tr->squeeze_challenge(tr->user, ch);
    const HF v = hf_from_abi(ch);
    // ---- Lagrange denominators of every set in one inversion; interpolants R_ij; numerators
    {
        std::vector<HF> dens;
        for (auto& rs : sets)
            for (uint32_t a : rs.points) {
                HF d = hone();
                for (uint32_t b : rs.points) if (b != a) d = hmul(d, hsub(upts[a], upts[b]));
                dens.push_back(d);
            }
        hf_batch_invert(dens);
        size_t o = 0;
        for (auto& rs : sets) { rs.inv_den.assign(dens.begin() + o, dens.begin() + o + rs.points.size()); o += rs.points.size(); }
    }
    void *d_num, *d_quot, *d_hx, *d_lx, *d_com;
    ZK_TRY(ctx->get_scratch("sp_num", nsets * n * 32, &d_num));
    ZK_TRY(ctx->get_scratch("sp_quot", (size_t)np * n * 32, &d_quot));
    ZK_TRY(ctx->get_scratch("sp_hx", n * 32, &d_hx));
    ZK_TRY(ctx->get_scratch("sp_lx", n * 32, &d_lx));
    ZK_TRY(ctx->get_scratch("sp_com", 96, &d_com));
    d_com = (char*)ctx->h_pinned + 2048;   // straight into pinned host memory: no read-back copy
    for (size_t si = 0; si < nsets; ++si) {
        RotSet& rs = sets[si];
        const size_t m = rs.points.size();
        // basis_i(X) = inv_den_i * prod_{j != i} (X - x_j), coefficients ascending
        std::vector<std::vector<HF>> basis(m);
        for (size_t i = 0; i < m; ++i) {
            std::vector<HF> num(1, hone());
            for (size_t j = 0; j < m; ++j) {
                if (j == i) continue;
                std::vector<HF> nx(num.size() + 1, hzero());
                for (size_t d = 0; d < num.size(); ++d) { nx[d + 1] = hadd(nx[d + 1], num[d]); nx[d] = hsub(nx[d], hmul(upts[rs.points[j]], num[d])); }
                num.swap(nx);
            }
            for (auto& c : num) c = hmul(c, rs.inv_den[i]);
            basis[i] = num;
        }
        std::vector<HF> low(m, hzero());
        std::vector<fe32> coeffs;
        std::vector<const void*> ptrs;
        HF yp = hone();
        for (size_t ci = 0; ci < rs.commits.size(); ++ci) {
            std::vector<HF> r(m, hzero());
            for (size_t i = 0; i < m; ++i)
                for (size_t d = 0; d < m; ++d) r[d] = hadd(r[d], hmul(rs.evals[ci][i], basis[i][d]));
            for (size_t d = 0; d < m; ++d) low[d] = hadd(low[d], hmul(yp, r[d]));
            coeffs.push_back(hf_raw(yp));
            ptrs.push_back(d_polys[rs.commits[ci]]);
            yp = hmul(yp, y);
        }
        std::vector<fe32> low_abi(m);
        for (size_t d = 0; d < m; ++d) low_abi[d] = hf_abi(low[d]);
        if (sharded && m > nl) { set_error("zkhip_shplonk_open: a rotation set of %zu points exceeds the rank's %zu rows", m, nl); return ZKHIP_EINVAL; }
        for (auto& q_ : ptrs) q_ = (const char*)q_ + lo_b;
        ZK_TRY(lincomb_raw(ctx, nl, ptrs.data(), ptrs.size(), coeffs.data(), low_abi.data(), lo == 0 ? m : 0, (char*)d_num + si * n * 32 + lo_b));
        rs.low = low;
    }
    // ---- h(X) = sum_i v^i g_i(X) / Z_i(X) by partial fractions, g_i = the set's numerator: sum_i v^i sum_{r in set i} w_ir g_i / (X - r).
    // The division's quotient is LINEAR in its numerator, so the terms are grouped by ROOT (round 4; until then one division per (set, root)
    // pair — 11 of them for the aggregation circuit's four sets over four points — and one 11-term combination):
    //     h(X) = sum_r [ sum_{i : r in set i} v^i w_ir g_i(X) ] / (X - r)
    // one division per DISTINCT point, all in one pass; a point of a single set divides that set's numerator directly and takes its
    // weight in the final sum.
    std::vector<HF> vpow(nsets);
    {
        HF acc = hone();
        for (size_t i = 0; i < nsets; ++i) { vpow[i] = acc; acc = hmul(acc, v); }
    }
    {
        std::vector<KdEntry> ents;
        std::vector<fe32> weights;
        std::vector<const void*> quots;
        for (uint32_t a = 0; a < np; ++a) {
            std::vector<const void*> ptrs;
            std::vector<fe32> cf;
            HF w1 = hone();
            for (size_t si = 0; si < nsets; ++si)
                for (size_t i = 0; i < sets[si].points.size(); ++i)
                    if (sets[si].points[i] == a) {
                        ptrs.push_back((const char*)d_num + si * n * 32 + lo_b);
                        w1 = hmul(vpow[si], sets[si].inv_den[i]);
                        cf.push_back(hf_raw(w1));
                    }
            if (ptrs.empty()) continue;   // cannot happen: every unique point comes from a query
            KdEntry e;
            e.dst = (uint32_t*)((char*)d_quot + quots.size() * n * 32 + lo_b);
            e.r = hf_raw(upts[a]);
            if (ptrs.size() == 1) {
                e.src = (const uint32_t*)ptrs[0];
                weights.push_back(hf_raw(w1));
            } else {
                ZK_TRY(lincomb_raw(ctx, nl, ptrs.data(), ptrs.size(), cf.data(), nullptr, 0, e.dst));
                e.src = e.dst;
                weights.push_back(hf_raw(hone()));
            }
            ents.push_back(e);
            quots.push_back(e.dst);
        }
        { ProfScope ps(ctx, "kate_division"); ZK_TRY(divide_round(ctx, nl, ents, sharded)); }
        ZK_TRY(lincomb_raw(ctx, nl, quots.data(), quots.size(), weights.data(), nullptr, 0, (char*)d_hx + lo_b));
    }
    uint8_t bytes[32];
    {
        const void* col[1] = {d_hx};
        ZK_TRY(zkhip_msm_g1_batch_device(ctx, srs, col, 1, n, d_com));
        ZK_TRY(zkhip_commitments_read(ctx, d_com, 1, h1_xy, bytes));
        tr->write_point(tr->user, bytes, h1_xy);
    }
    // ---- u; L(X) = sum_i v^i z_i sum_j y^j (P_ij(X) - R_ij(u)) - Z_T(u) h(X), all scaled by 1 / z_0; h'(X) = L(X) / (X - u)
    tr->squeeze_challenge(tr->user, ch);
    const HF u = hf_from_abi(ch);
    std::vector<HF> zdiff(nsets);
    HF zt = hone();
    for (uint32_t a = 0; a < np; ++a) zt = hmul(zt, hsub(u, upts[a]));
    for (size_t si = 0; si < nsets; ++si) {
        HF z = hone();
        for (uint32_t a = 0; a < np; ++a) {
            bool in_set = false;
            for (uint32_t b : sets[si].points) in_set |= b == a;
            if (!in_set) z = hmul(z, hsub(u, upts[a]));
        }
        zdiff[si] = z;
    }
    if (hf_is_zero(zdiff[0])) { set_error("zkhip_shplonk_open: the challenge u hit an opening point"); return ZKHIP_EINVAL; }
    std::vector<HF> one_inv(1, zdiff[0]);
    hf_batch_invert(one_inv);
    const HF inv0 = one_inv[0];
    {
        // sum_j y^j P_ij(X) = g_i(X) + R_i(X): the sets' numerators of the first round are still in d_num, so L(X) is a combination of
        // nsets + 1 polynomials and a correction of max |set| low coefficients (until round 4: every queried polynomial read again)
        std::vector<fe32> coeffs;
        std::vector<const void*> ptrs;
        size_t maxm = 1;
        for (auto& rs : sets) maxm = std::max(maxm, rs.low.size());
        std::vector<HF> lowv(maxm, hzero());   // out = sum c p - lowv: lowv[d] = sum_i scale_i ([d = 0] R_i(u) - R_i[d])
        for (size_t si = 0; si < nsets; ++si) {
            const HF scale = hmul(hmul(vpow[si], zdiff[si]), inv0);
            HF ru = hzero();   // R_i(u) by Horner
            for (size_t d = sets[si].low.size(); d-- > 0;) ru = hadd(hmul(ru, u), sets[si].low[d]);
            lowv[0] = hadd(lowv[0], hmul(scale, ru));
            for (size_t d = 0; d < sets[si].low.size(); ++d) lowv[d] = hsub(lowv[d], hmul(scale, sets[si].low[d]));
            coeffs.push_back(hf_raw(scale));
            ptrs.push_back((const char*)d_num + si * n * 32 + lo_b);
        }
        coeffs.push_back(hf_raw(hsub(hzero(), hmul(zt, inv0))));
        ptrs.push_back((const char*)d_hx + lo_b);
        std::vector<fe32> low_abi(maxm);
        for (size_t d = 0; d < maxm; ++d) low_abi[d] = hf_abi(lowv[d]);
        ZK_TRY(lincomb_raw(ctx, nl, ptrs.data(), ptrs.size(), coeffs.data(), low_abi.data(), lo == 0 ? maxm : 0, (char*)d_lx + lo_b));
        std::vector<KdEntry> ents(1);
        ents[0].src = (const uint32_t*)((char*)d_lx + lo_b); ents[0].dst = (uint32_t*)((char*)d_lx + lo_b); ents[0].r = hf_raw(u);
        { ProfScope ps(ctx, "kate_division"); ZK_TRY(divide_round(ctx, nl, ents, sharded)); }
        const void* col[1] = {d_lx};
        ZK_TRY(zkhip_msm_g1_batch_device(ctx, srs, col, 1, n, d_com));
        ZK_TRY(zkhip_commitments_read(ctx, d_com, 1, h2_xy, bytes));
        tr->write_point(tr->user, bytes, h2_xy);
    }
    return ZKHIP_OK;
}

extern "C" {

int zkhip_shplonk_open(zkhip_ctx* ctx, const zkhip_srs* srs, size_t n, const void* const* d_polys, size_t npolys, const uint32_t* query_poly,
                       const uint64_t* query_points, const uint64_t* query_evals, size_t nq, const zk_transcript* tr, uint64_t h1_xy[8],
                       uint64_t h2_xy[8]) {
    return zk::shplonk_open(ctx, srs, n, d_polys, npolys, query_poly, query_points, query_evals, nq, tr, h1_xy, h2_xy, false);
}

int zkhip_linear_combination_device(zkhip_ctx* ctx, size_t n, const void* const* d_polys, size_t npolys, const uint64_t* coeffs,
                                    const uint64_t* low, size_t nlow, void* d_out) {
    if (!ctx || !d_out || (npolys && (!d_polys || !coeffs)) || (nlow && !low)) { set_error("zkhip_linear_combination_device: null argument"); return ZKHIP_EINVAL; }
    if (nlow > n) { set_error("zkhip_linear_combination_device: nlow = %zu exceeds n = %zu", nlow, n); return ZKHIP_EINVAL; }
    if (n == 0) return ZKHIP_OK;
    std::vector<fe32> cf(npolys ? npolys : 1), lw(nlow ? nlow : 1);
    for (size_t j = 0; j < npolys; ++j) cf[j] = abi_to_raw(coeffs + 4 * j);
    for (size_t j = 0; j < nlow; ++j) lw[j] = mem_load(low + 4 * j);
    return lincomb_raw(ctx, n, d_polys, npolys, cf.data(), lw.data(), nlow, d_out);
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
