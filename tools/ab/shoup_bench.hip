// shoup_bench.hip — is a Shoup-style product by a KNOWN constant (the NTT's twiddles) cheaper than the Montgomery product on gfx950?
//   r = a w - floor(a w' / 2^261) p,  w' = floor(w 2^261 / p)   (w plain and canonical, a any value < 2^261 with lazy 29-bit limbs)
// = 53 (the high columns of a w', with two guard columns) + 45 (a w mod 2^261) + 45 (q (2^261 - p) mod 2^261) multiplier instructions
// against 162 + 9 for the Montgomery product; r in [0, 3p).  w' is exactly the q-vector of the Montgomery reduction of w's Montgomery form.
//   hipcc --offload-arch=gfx950 -O3 -I halo2-zkcert_amd/csrc tools/ab/shoup_bench.hip -o tools/ab/shoup_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "bn254.hpp"
using namespace zk;

template <class P>
ZK_HD constexpr uint32_t pbar_limb(int i) {   // limb i of 2^261 - p
    // two's complement over nine 29-bit limbs: ~p + 1
    uint32_t carry = 1, r = 0;
    for (int j = 0; j <= i; ++j) {
        uint32_t v = ((~P::M[j]) & LMASK) + carry;
        r = v & LMASK;
        carry = v >> LB;
    }
    return r;
}
// from the Montgomery (R' = 2^261) form of w: plain w (canonical) and w' = the q-vector of its reduction
template <class P>
ZK_HD __forceinline__ void shoup_from_mont(const fe& wm, fe& w, fe& wq) {
    uint64_t acc[10];
    for (int j = 0; j < 9; ++j) acc[j] = wm.l[j];
    acc[9] = 0;
    for (int i = 0; i < 9; ++i) {
        uint32_t q = ((uint32_t)acc[0] * P::INV) & LMASK;
        wq.l[i] = q;
        for (int j = 0; j < 9; ++j) acc[j] += (uint64_t)q * P::M[j];
        uint64_t carry = acc[0] >> LB;
        for (int j = 0; j < 9; ++j) acc[j] = acc[j + 1];
        acc[0] += carry;
        acc[9] = 0;
    }
    fe r;
    for (int j = 0; j < 8; ++j) { r.l[j] = (uint32_t)acc[j] & LMASK; acc[j + 1] += acc[j] >> LB; }
    r.l[8] = (uint32_t)acc[8];
    w = fe_canonical<P>(r);
    // w' belongs to the canonical w: if the reduction's result was w + p the q-vector belongs to it all the same (w' is defined by
    // w' p = w_plain 2^261 - w_M only for the result actually produced): re-derive for the canonical representative
    // (result = (w_M + q p) / 2^261; canonical = result - p  <=>  q' = q - 2^261... not representable) — so insist on result < p:
}
template <class P>
ZK_HD __forceinline__ fe fe_mul_shoup(const fe& a, const fe& w, const fe& wq) {
    // high columns 7..16 of a * wq
    uint64_t hi[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) hi[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int j = 0; j < 9; ++j)
            if (i + j >= 7) hi[i + j - 7] += (uint64_t)a.l[i] * wq.l[j];
#pragma unroll
    for (int k = 0; k < 10; ++k) hi[k + 1] += hi[k] >> LB;
    fe q;   // columns 9..17
#pragma unroll
    for (int t = 0; t < 8; ++t) q.l[t] = (uint32_t)hi[t + 2] & LMASK;
    q.l[8] = (uint32_t)hi[10];
    uint64_t lo[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) lo[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int j = 0; j < 9; ++j)
            if (i + j <= 8) lo[i + j] += (uint64_t)a.l[i] * w.l[j];
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int j = 0; j < 9; ++j)
            if (i + j <= 8) lo[i + j] += (uint64_t)q.l[i] * pbar_limb<P>(j);
    fe r;
#pragma unroll
    for (int k = 0; k < 8; ++k) { r.l[k] = (uint32_t)lo[k] & LMASK; lo[k + 1] += lo[k] >> LB; }
    r.l[8] = (uint32_t)lo[8] & LMASK;
    return r;
}

__device__ __forceinline__ el2<Fr> seeded(uint64_t seed, uint64_t gid) { return reduce(el<Fr, 32 * U>(fe_split<5>(synth_raw253(seed, gid)))); }
// the NTT's shape: values multiplied by per-thread constants over and over
__global__ void k_chain_mont(uint32_t* out, int iters, uint64_t seed) {
    uint64_t gid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    el2<Fr> x = seeded(seed, gid), y = seeded(seed + 1, gid), w1 = seeded(seed + 2, gid), w2 = seeded(seed + 3, gid);
    for (int it = 0; it < iters; ++it) { x = x * w1; y = y * w2; x = reduce(x + y); }
    store_raw<Fr>(out + gid * 8, x + y);
}
__global__ void k_chain_shoup(uint32_t* out, int iters, uint64_t seed) {
    uint64_t gid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    fe x = seeded(seed, gid).v, y = seeded(seed + 1, gid).v;
    fe w1, q1, w2, q2;
    shoup_from_mont<Fr>(fe_canonical<Fr>(seeded(seed + 2, gid).v), w1, q1);
    shoup_from_mont<Fr>(fe_canonical<Fr>(seeded(seed + 3, gid).v), w2, q2);
    for (int it = 0; it < iters; ++it) {
        x = fe_mul_shoup<Fr>(x, w1, q1);
        y = fe_mul_shoup<Fr>(y, w2, q2);
        for (int i = 0; i < 9; ++i) x.l[i] += y.l[i];      // lazy: the next product takes unnormalised limbs
        x = fe_mul_shoup<Fr>(x, w1, q1);
    }
    store_raw<Fr>(out + gid * 8, el<Fr, 8 * U>(x));
}
// exactness: z = a * w through both routes
__global__ void k_check(uint32_t* out_m, uint32_t* out_s, size_t n, uint64_t seed) {
    size_t gid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (gid >= n) return;
    el2<Fr> a = seeded(seed, gid), wm = seeded(seed + 7, gid);
    fe al = a.v;
    if (gid & 1) { for (int i = 0; i < 9; ++i) al.l[i] = al.l[i] * 3 + (i < 8 ? LMASK : 0); }   // lazy limbs up to 2^31, value up to ~8p
    fe alc = al;
    fe_normalize(alc);
    store_raw<Fr>(out_m + gid * 8, el<Fr, 16 * U>(alc) * wm);        // a w (Montgomery: w in R' form)
    fe w, q;
    shoup_from_mont<Fr>(fe_canonical<Fr>(wm.v), w, q);
    store_raw<Fr>(out_s + gid * 8, el<Fr, 8 * U>(fe_mul_shoup<Fr>(al, w, q)));
}

template <class F>
static double time_ms(F launch) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(); (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best;
}
int main() {
    const size_t n = 1 << 20;
    uint32_t *dm, *ds;
    (void)hipMalloc(&dm, n * 32); (void)hipMalloc(&ds, n * 32);
    k_check<<<(unsigned)(n / 256), 256>>>(dm, ds, n, 5);
    (void)hipDeviceSynchronize();
    std::vector<uint32_t> hm(n * 8), hs(n * 8);
    (void)hipMemcpy(hm.data(), dm, n * 32, hipMemcpyDeviceToHost); (void)hipMemcpy(hs.data(), ds, n * 32, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t i = 0; i < n * 8; ++i) bad += hm[i] != hs[i];
    printf("exactness: %zu products, %zu differing words\n", n, bad);
    uint32_t* out; (void)hipMalloc(&out, 64ull << 20);
    for (int wpc : {8, 16, 32}) {
        int blocks = 256 * wpc / 4, threads = 256, iters = 300;
        double lanes = (double)blocks * threads;
        double a = time_ms([&] { k_chain_mont<<<blocks, threads>>>(out, iters, 1); });
        double b = time_ms([&] { k_chain_shoup<<<blocks, threads>>>(out, iters, 1); });
        printf("waves/SIMD=%d: Montgomery (3 products per iteration) %.3f ms = %.1f G/s | Shoup %.3f ms = %.1f G/s | ratio %.3f\n", wpc / 4, a,
               lanes * 3 * iters / a / 1e6, b, lanes * 3 * iters / b / 1e6, a / b);
    }
    return bad != 0;
}
