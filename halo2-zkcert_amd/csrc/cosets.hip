// cosets.hip — the quotient polynomial on quotient_poly_degree cosets of the size-n domain instead of the 2^extended_k extended domain.
//
// halo2's evaluate_h works on the extended domain of 2^extended_k >= (d - 1) n points, d = cs.degree()
// [UPSTREAM-RECALL halo2_proofs src/poly/domain.rs EvaluationDomain::new / coeff_to_extended / extended_to_coeff / divide_by_vanishing_poly;
//  crate pinned at /root/reference/Cargo.lock:1320-1322].  The extended domain g <w_ext> is the union of E = 2^(extended_k - k) cosets
// s_r H, s_r = g w_ext^r, of the size-n domain H = <w>; the quotient h has fewer than q n coefficients (q = d - 1, its number of
// pieces), so its values on ANY q of those cosets determine it.  When q < E (d = 4: 3 of 4 cosets, d = 6: 5 of 8) the prover
// therefore
//   * evaluates every column on q cosets only: block r = NTT_w(a_t s_r^t)            (q size-n transforms instead of one of size E n),
//   * sweeps q n rows (sweep.hip, coset addressing: a rotation stays inside its block),
//   * and goes back coset by coset: U_r = s_r^-t iNTT_raw(N|coset r)[t] = n (c_r - 1) sum_j c_r^j h_j[t], c_r = s_r^n (the numerator
//     N = h (X^n - 1) is constant c_r - 1 times h on coset r), so the pieces are h_j = sum_r M[j][r] U_r with
//     M = V^-1 diag(1 / (n (c_r - 1))), V[r][j] = c_r^j — a q x q host-side inversion and q linear combinations of q columns.
// The result is the same polynomial h, hence the same pieces, commitments and proof bytes as the extended-domain path (which remains
// the one the natural-order C-ABI entry points expose and is what the Python schedule and the oracle run: the tests compare the two).
// The proving key's fixed / sigma / l_0 / l_last / l_active columns are brought into the same q-block layout once per key_id.
#include <memory>

#include "common.hpp"
#include "hostfield.hpp"
using namespace zk;

struct zk::CosetPlan {
    uint32_t k = 0, q = 0;
    alignas(16) uint64_t omega_abi[4], omega_inv_abi[4];   // read with 16-byte loads (mem_load)
    std::vector<uint64_t> shifts_abi;   // q x 4: s_r
    std::vector<uint64_t> minv_abi;     // [j][r] x 4
    void* d_pre = nullptr;              // q tables of n: s_r^t   (raw R' form)
    void* d_post = nullptr;             // q tables of n: s_r^-t
    std::map<uint64_t, KeyCosets> keys;
};

uint32_t zk::coset_plan_q(const CosetPlan* p) { return p->q; }
void zk::coset_sweep_view(const CosetPlan* p, SweepCosets* out) {
    out->q = p->q;
    out->shifts_abi = p->shifts_abi.data();
    out->omega_abi = p->omega_abi;
}

static inline void put_abi(uint64_t* out, const HF& a) { memcpy(out, a.w, 32); }   // HF is the ABI form; no alignment assumed

// A device buffer that becomes one of the context's persistent (named) buffers only when its owner says the object it belongs to is
// complete: until keep() the destructor frees it, so a plan / key that fails half-way leaves nothing behind, and a retry cannot
// overwrite (leak) an earlier allocation of the same name.
struct PendingBuffer {
    zkhip_ctx* ctx;
    std::string name;
    void* ptr = nullptr;
    PendingBuffer(zkhip_ctx* c, std::string nm) : ctx(c), name(std::move(nm)) {}
    PendingBuffer(const PendingBuffer&) = delete;
    int alloc(size_t bytes) {
        auto old = ctx->persistent.find(name);   // a stale entry (no host-side object refers to it, or we would not be here)
        if (old != ctx->persistent.end()) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(old->second); ctx->persistent.erase(old); }
        hipError_t e = zk::dev_malloc((void**)&ptr, bytes);
        if (e != hipSuccess) { (void)hipGetLastError(); ptr = nullptr; set_error("hipMalloc(%zu) for %s failed: %s", bytes, name.c_str(), hipGetErrorString(e)); return ZKHIP_ENOMEM; }
        return ZKHIP_OK;
    }
    void keep() { if (ptr) ctx->persistent[name] = ptr; ptr = nullptr; }   // freed with the context (or by zkhip_key_release)
    ~PendingBuffer() { if (ptr) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(ptr); } }
};

int zk::coset_plan(zkhip_ctx* ctx, const zkhip_domain* d, const CosetPlan** out) {
    if (!ctx || !d || !out) { set_error("coset_plan: null argument"); return ZKHIP_EINVAL; }
    const uint32_t k = zkhip_domain_k(d), ek = zkhip_domain_extended_k(d), q = zkhip_domain_quotient_poly_degree(d);
    uint64_t omega[4], ext_omega[4], g[4];
    zkhip_domain_constants(d, omega, ext_omega, g);
    char name[160];
    snprintf(name, sizeof name, "coset_plan:%u:%u:%u:%016llx%016llx%016llx%016llx", k, ek, q, (unsigned long long)g[3], (unsigned long long)g[2],
             (unsigned long long)g[1], (unsigned long long)g[0]);
    auto it = ctx->host_objects.find(name);
    if (it != ctx->host_objects.end()) { *out = (const CosetPlan*)it->second.get(); return ZKHIP_OK; }
    if (q < 2 || q >= (1u << (ek - k))) { set_error("coset_plan: %u cosets do not save rows over extended_k = %u", q, ek); return ZKHIP_EINVAL; }
    auto plan = std::make_shared<CosetPlan>();
    plan->k = k; plan->q = q;
    const size_t n = (size_t)1 << k;
    memcpy(plan->omega_abi, omega, 32);
    const HF w = hf_from_abi(omega), we = hf_from_abi(ext_omega), hg = hf_from_abi(g);
    put_abi(plan->omega_inv_abi, hf_invert(w));
    plan->shifts_abi.resize(4 * q);
    plan->minv_abi.resize(4 * (size_t)q * q);
    PendingBuffer b_pre(ctx, std::string(name) + ":pre"), b_post(ctx, std::string(name) + ":post");
    ZK_TRY(b_pre.alloc((size_t)q * n * 32));
    ZK_TRY(b_post.alloc((size_t)q * n * 32));
    plan->d_pre = b_pre.ptr;
    plan->d_post = b_post.ptr;
    std::vector<HF> c(q), tinv(q);
    HF s = hg;
    HF ninv = hf_invert(hf_from_fe32(to_abi(from_u64<Fr>((uint64_t)n))));
    for (uint32_t r = 0; r < q; ++r) {
        put_abi(plan->shifts_abi.data() + 4 * r, s);
        uint64_t sinv[4];
        put_abi(sinv, hf_invert(s));
        ZK_TRY(power_table(ctx, plan->shifts_abi.data() + 4 * r, n, (char*)plan->d_pre + (size_t)r * n * 32));
        ZK_TRY(power_table(ctx, sinv, n, (char*)plan->d_post + (size_t)r * n * 32));
        c[r] = hpow(s, n);
        HF t = hsub(c[r], hone());
        if (hf_is_zero(t)) { set_error("coset_plan: coset %u lies on the domain", r); return ZKHIP_EINVAL; }
        tinv[r] = hmul(hf_invert(t), ninv);
        s = hmul(s, we);
    }
    // V[r][j] = c_r^j; Gauss-Jordan on [V | I]
    std::vector<HF> a((size_t)q * 2 * q);
    auto at = [&](uint32_t r, uint32_t col) -> HF& { return a[(size_t)r * 2 * q + col]; };
    for (uint32_t r = 0; r < q; ++r) {
        HF p = hone();
        for (uint32_t j = 0; j < q; ++j) { at(r, j) = p; p = hmul(p, c[r]); at(r, q + j) = j == r ? hone() : hzero(); }
    }
    for (uint32_t col = 0; col < q; ++col) {
        uint32_t piv = col;
        while (piv < q && hf_is_zero(at(piv, col))) ++piv;
        if (piv == q) { set_error("coset_plan: singular coset system"); return ZKHIP_EINVAL; }
        if (piv != col) for (uint32_t j = 0; j < 2 * q; ++j) std::swap(at(piv, j), at(col, j));
        HF iv = hf_invert(at(col, col));
        for (uint32_t j = 0; j < 2 * q; ++j) at(col, j) = hmul(at(col, j), iv);
        for (uint32_t r = 0; r < q; ++r) {
            if (r == col || hf_is_zero(at(r, col))) continue;
            HF f = at(r, col);
            for (uint32_t j = 0; j < 2 * q; ++j) at(r, j) = hsub(at(r, j), hmul(f, at(col, j)));
        }
    }
    for (uint32_t j = 0; j < q; ++j)
        for (uint32_t r = 0; r < q; ++r) put_abi(plan->minv_abi.data() + 4 * ((size_t)j * q + r), hmul(at(j, q + r), tinv[r]));
    ZK_HIP(hipStreamSynchronize(ctx->stream));   // one-time: other streams of the context read the tables too
    b_pre.keep();
    b_post.keep();
    ctx->host_objects[name] = plan;
    *out = plan.get();
    return ZKHIP_OK;
}

// dsts[j]: q n elements, block r = srcs[j] (n coefficients) evaluated on s_r H.  One batch of q npolys size-n transforms.
int zk::coeff_to_cosets(zkhip_ctx* ctx, const CosetPlan* p, const void* const* srcs, void* const* dsts, size_t npolys) {
    if (!ctx || !p || !srcs || !dsts) { set_error("coeff_to_cosets: null argument"); return ZKHIP_EINVAL; }
    if (!npolys) return ZKHIP_OK;
    const size_t NB = ((size_t)1 << p->k) * 32;
    std::vector<const void*> s(p->q * npolys);
    std::vector<void*> d(p->q * npolys);
    for (uint32_t r = 0; r < p->q; ++r)
        for (size_t j = 0; j < npolys; ++j) { s[r * npolys + j] = srcs[j]; d[r * npolys + j] = (char*)dsts[j] + r * NB; }
    return ntt_tabled(ctx, s.data(), d.data(), s.size(), p->omega_abi, p->k, p->d_pre, nullptr, (uint32_t)npolys, (size_t)1 << p->k);
}

// d_vals: the sweep's output (q blocks of n numerator values; overwritten), d_pieces: q n coefficients, piece j at j n
int zk::cosets_to_pieces(zkhip_ctx* ctx, const CosetPlan* p, void* d_vals, void* d_pieces) {
    if (!ctx || !p || !d_vals || !d_pieces) { set_error("cosets_to_pieces: null argument"); return ZKHIP_EINVAL; }
    const size_t n = (size_t)1 << p->k, NB = n * 32;
    std::vector<void*> u(p->q);
    for (uint32_t r = 0; r < p->q; ++r) u[r] = (char*)d_vals + r * NB;
    ZK_TRY(ntt_tabled(ctx, (const void* const*)u.data(), u.data(), p->q, p->omega_inv_abi, p->k, nullptr, p->d_post, 1, n));
    for (uint32_t j = 0; j < p->q; ++j)
        ZK_TRY(zkhip_linear_combination_device(ctx, n, (const void* const*)u.data(), p->q, p->minv_abi.data() + 4 * (size_t)j * p->q, nullptr, 0,
                                               (char*)d_pieces + j * NB));
    return ZKHIP_OK;
}

// The same in two steps for the row-sharded form (prover.hip): the inverse transform of SOME blocks (their owner's), and the combination
// restricted to a row range (pointwise in the rows).
int zk::cosets_inverse_blocks(zkhip_ctx* ctx, const CosetPlan* p, void* d_vals, const uint32_t* blocks, size_t nblocks) {
    if (!ctx || !p || !d_vals || (nblocks && !blocks)) { set_error("cosets_inverse_blocks: null argument"); return ZKHIP_EINVAL; }
    const size_t n = (size_t)1 << p->k, NB = n * 32;
    for (size_t i = 0; i < nblocks; ++i) {
        if (blocks[i] >= p->q) { set_error("cosets_inverse_blocks: block %u of %u", blocks[i], p->q); return ZKHIP_EINVAL; }
        void* u[1] = {(char*)d_vals + blocks[i] * NB};
        ZK_TRY(ntt_tabled(ctx, (const void* const*)u, u, 1, p->omega_inv_abi, p->k, nullptr, (const char*)p->d_post + blocks[i] * NB, 1, n));
    }
    return ZKHIP_OK;
}
int zk::cosets_combine_range(zkhip_ctx* ctx, const CosetPlan* p, const void* d_vals, void* d_pieces, size_t first_row, size_t count) {
    if (!ctx || !p || !d_vals || !d_pieces) { set_error("cosets_combine_range: null argument"); return ZKHIP_EINVAL; }
    const size_t n = (size_t)1 << p->k, NB = n * 32;
    if (first_row + count > n) { set_error("cosets_combine_range: rows [%zu, %zu) of %zu", first_row, first_row + count, n); return ZKHIP_EINVAL; }
    std::vector<const void*> u(p->q);
    for (uint32_t r = 0; r < p->q; ++r) u[r] = (const char*)d_vals + r * NB + first_row * 32;
    for (uint32_t j = 0; j < p->q; ++j)
        ZK_TRY(zkhip_linear_combination_device(ctx, count, u.data(), p->q, p->minv_abi.data() + 4 * (size_t)j * p->q, nullptr, 0,
                                               (char*)d_pieces + j * NB + first_row * 32));
    return ZKHIP_OK;
}

// out[i] = (lo <= i < hi) ? 1 : 0 in the ABI form: the Lagrange-basis vectors of l_0, l_last and l_active_row
__global__ void k_indicator(uint32_t* out, size_t n, size_t lo, size_t hi, fe32 one_abi) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe32 z;
    for (int w = 0; w < 8; ++w) z.w[w] = 0;
    mem_store(out + i * 8, (i >= lo && i < hi) ? one_abi : z);
}

int zk::key_cosets(zkhip_ctx* ctx, const CosetPlan* cp, const zk_proving_key* pk, const KeyCosets** out) {
    if (!ctx || !cp || !pk || !out) { set_error("key_cosets: null argument"); return ZKHIP_EINVAL; }
    if (!pk->key_id) { set_error("key_cosets: the proving key has no key_id"); return ZKHIP_EINVAL; }
    CosetPlan* p = const_cast<CosetPlan*>(cp);   // the plan is the context's own cache object
    auto it = p->keys.find(pk->key_id);
    if (it != p->keys.end()) { *out = &it->second; return ZKHIP_OK; }
    const uint32_t F = pk->n_fixed, P = pk->n_perm_columns;
    if ((F && !pk->fixed_coeff) || (P && !pk->sigma_coeff)) { set_error("key_cosets: fixed_coeff / sigma_coeff missing"); return ZKHIP_EINVAL; }
    const size_t n = (size_t)1 << p->k, NB = n * 32, CB = p->q * NB;
    char name[96];
    snprintf(name, sizeof name, "key_cosets:%llx:%u:%u", (unsigned long long)pk->key_id, p->k, p->q);
    PendingBuffer b_key(ctx, name);
    ZK_TRY(b_key.alloc((size_t)(F + P + 3) * CB));
    void* base = b_key.ptr;
    KeyCosets kc;
    std::vector<const void*> src;
    std::vector<void*> dst;
    for (uint32_t i = 0; i < F + P; ++i) {
        void* d = (char*)base + (size_t)i * CB;
        (i < F ? kc.fixed : kc.sigma).push_back(d);
        src.push_back(i < F ? pk->fixed_coeff[i] : pk->sigma_coeff[i - F]);
        dst.push_back(d);
    }
    // in batches of four polynomials: the transform's pass buffer is sized by the batch, and one batch of all F + P columns (4.7 GiB at
    // k = 22) would stay in the context for good as the largest scratch buffer it ever needed — first-touched once, never used again
    for (size_t i0 = 0; i0 < src.size(); i0 += 4)
        ZK_TRY(coeff_to_cosets(ctx, p, src.data() + i0, dst.data() + i0, std::min<size_t>(4, src.size() - i0)));
    // l_0, l_last (row n - blinding_factors - 1), l_active_row = 1 - l_last - sum of the blinding rows' basis polynomials:
    // Lagrange indicator vectors -> coefficients -> cosets
    void* d_lag;
    ZK_TRY(ctx->get_scratch("key_cosets_lagrange", 3 * NB, &d_lag));
    const size_t last = n - pk->blinding_factors - 1;
    const fe32 one_abi = hf_abi(hone());
    const size_t rng[3][2] = {{0, 1}, {last, last + 1}, {0, last}};
    void* lag[3];
    void* l_dst[3];
    for (int i = 0; i < 3; ++i) {
        lag[i] = (char*)d_lag + i * NB;
        l_dst[i] = (char*)base + (size_t)(F + P + i) * CB;
        hipLaunchKernelGGL(k_indicator, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, (uint32_t*)lag[i], n, rng[i][0], rng[i][1], one_abi);
    }
    ZK_LAUNCH_CHECK();
    ZK_TRY(zkhip_lagrange_to_coeff_device(ctx, pk->domain, lag, 3));
    ZK_TRY(coeff_to_cosets(ctx, p, (const void* const*)lag, l_dst, 3));
    kc.l0 = l_dst[0]; kc.l_last = l_dst[1]; kc.l_active = l_dst[2];
    ZK_HIP(hipStreamSynchronize(ctx->stream));   // one-time; later proofs may read these from any stream of the context
    b_key.keep();
    auto ins = p->keys.emplace(pk->key_id, std::move(kc));
    *out = &ins.first->second;
    return ZKHIP_OK;
}

void zk::coset_forget_key(zkhip_ctx* ctx, uint64_t key_id) {
    for (auto& kv : ctx->host_objects)
        if (kv.first.compare(0, 11, "coset_plan:") == 0) static_cast<CosetPlan*>(kv.second.get())->keys.erase(key_id);
}

// ------------------------------------------------------------------ C ABI
extern "C" {
// 1 if zkhip_create_proof_ex will evaluate this key's quotient on cosets of the size-n domain (so the key's extended-domain forms —
// fixed_cosets, sigma_cosets, l0, l_last, l_active_row — may be NULL), 0 if it will use the extended domain and needs them.  The same
// predicate the prover evaluates: a binding asks instead of guessing (the Rust shim decided on its own).
int zkhip_coset_quotient_applies(const zkhip_ctx* ctx, const zk_proving_key* pk) {
    if (!ctx || !pk || !pk->domain) return 0;
    const uint32_t k = zkhip_domain_k(pk->domain), ek = zkhip_domain_extended_k(pk->domain), qd = zkhip_domain_quotient_poly_degree(pk->domain);
    return ctx->opt.coset_quotient != 0 && pk->key_id != 0 && qd >= 2 && qd < (1u << (ek - k)) && (!pk->n_fixed || pk->fixed_coeff) &&
           (!pk->n_perm_columns || pk->sigma_coeff);
}
int zkhip_domain_cosets(zkhip_ctx* ctx, const zkhip_domain* dom, uint32_t* q, uint64_t* shifts) {
    const CosetPlan* p;
    ZK_TRY(coset_plan(ctx, dom, &p));
    if (q) *q = p->q;
    if (shifts) memcpy(shifts, p->shifts_abi.data(), (size_t)p->q * 32);
    return ZKHIP_OK;
}
int zkhip_coeff_to_cosets_device(zkhip_ctx* ctx, const zkhip_domain* dom, const void* const* d_in, void* const* d_out, size_t npolys) {
    const CosetPlan* p;
    ZK_TRY(coset_plan(ctx, dom, &p));
    return coeff_to_cosets(ctx, p, d_in, d_out, npolys);
}
int zkhip_cosets_to_pieces_device(zkhip_ctx* ctx, const zkhip_domain* dom, void* d_vals, void* d_pieces) {
    const CosetPlan* p;
    ZK_TRY(coset_plan(ctx, dom, &p));
    return cosets_to_pieces(ctx, p, d_vals, d_pieces);
}
int zkhip_evaluate_h_cosets_device(zkhip_ctx* ctx, const zkhip_domain* dom, const zk_evalh_args* args, size_t first_row, size_t n_rows,
                                   void* d_out) {
    const CosetPlan* p;
    ZK_TRY(coset_plan(ctx, dom, &p));
    if (!args || args->k != p->k) { set_error("zkhip_evaluate_h_cosets_device: args do not match the domain"); return ZKHIP_EINVAL; }
    SweepCosets sc;
    coset_sweep_view(p, &sc);
    return evaluate_h_cosets(ctx, args, &sc, first_row, n_rows, d_out);
}
}
