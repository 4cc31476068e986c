// fp64_field.hip — the question VERDICT r2 item 3 asks: is a double-precision (52-bit limb, v_fma_f64 hi/lo split) Montgomery product
// faster on gfx950 than the 9 x 29-bit v_mad_u64_u32 product every hot kernel of this repo is issue-bound on?
//
// The scheme is the one of Emmart, Zheng, Weems, "Faster modular exponentiation using double precision floating point arithmetic on
// the GPU" (ARITH 2018), restated for BN254's base field: an element is 5 limbs of 52 bits, each held in a double; with the FP64 rounding
// mode set to round-toward-zero,
//      hi = fma(a, b, 2^104)                 = 2^104 + floor(a b / 2^52) 2^52      (exact: ulp of [2^104, 2^105) is 2^52)
//      lo = fma(a, b, (2^104 + 2^52) - hi)   = 2^52 + (a b mod 2^52)               (exact)
// and the bit patterns of hi / lo are (exponent | HI) / (exponent | LO), so column sums are taken with 64-bit INTEGER additions and the
// exponent fields are removed once per column as a compile-time constant.  Montgomery reduction (radix 2^52, R = 2^260) the same way.
//
//   hipcc --offload-arch=gfx950 -O3 -I halo2-zkcert_amd/csrc tools/fp64_field.hip -o tools/fp64_field
//   tools/fp64_field            # exactness on 2^20 random + edge operands against a host big-integer check, then throughput at 1/2/4/8 waves per SIMD
// Result and instruction mix: profiles/r03_fp64_field.md
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "hostfield.hpp"
using namespace zk;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// ---- Fq = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47 in 52-bit limbs; PINV = -p^-1 mod 2^52 (filled by the host)
struct DpConst { double p[5]; double pinv; };
__constant__ DpConst g_dp;

struct dpe { double l[5]; };   // value = sum l[i] 2^(52 i), every l[i] an integer in [0, 2^52); value < 2 p

__device__ __forceinline__ void set_round_toward_zero_f64() {
    // MODE register (hwreg 1), FP_ROUND bits [3:2] = double / half precision rounding: 3 = toward zero.  simm16 = id | offset << 6 | (size - 1) << 11
    // Through inline assembly on purpose: with the builtin, LLVM's mode-register pass knows the mode was changed and switches it BACK to
    // round-to-nearest in front of every FP64 instruction it emits (it assumes the default FP environment).
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3");
}
#define DP_C1 0x1p104
#define DP_C2 (0x1p104 + 0x1p52)
#define BITS_HI 0x4670000000000000ll   // bit pattern of 2^104
#define BITS_LO 0x4330000000000000ll   // bit pattern of 2^52
#define MASK52 0x000fffffffffffffll

__device__ __forceinline__ void mul_acc(double a, double b, long long& hi_col, long long& lo_col) {
    const double hi = __builtin_fma(a, b, DP_C1);
    const double sub = DP_C2 - hi;
    const double lo = __builtin_fma(a, b, sub);
    hi_col += __double_as_longlong(hi);
    lo_col += __double_as_longlong(lo);
}
// integer < 2^52 -> double: splice the bits under the exponent of 2^52, subtract 2^52 (exact)
__device__ __forceinline__ double to_double52(long long t) { return __longlong_as_double(t | BITS_LO) - 0x1p52; }

// a * b / 2^260 mod p, result < 2 p for a, b < 2 p (4 p < 2^260)
__device__ __forceinline__ dpe dp_mul(const dpe& a, const dpe& b) {
    long long c[11];
    // column k receives (number of lo terms, number of hi terms) from the product and from the five reduction rows: the exponent fields
    // those terms carry are known at compile time and go in as the columns' initial values
#pragma unroll
    for (int k = 0; k < 11; ++k) {
        int nlo = 0, nhi = 0;
        for (int i = 0; i < 5; ++i)
            for (int j = 0; j < 5; ++j) {
                if (i + j == k) nlo += 2;        // product term + reduction term q_i p_j
                if (i + j + 1 == k) nhi += 2;
            }
        c[k] = -(long long)((unsigned long long)nlo * (unsigned long long)BITS_LO + (unsigned long long)nhi * (unsigned long long)BITS_HI);
    }
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) mul_acc(a.l[i], b.l[j], c[i + j + 1], c[i + j]);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        // the low 52 bits of column i are final (the exponent fields only touch bits >= 52): q = that * pinv mod 2^52
        const double t = to_double52(c[i] & MASK52);
        const double hi = __builtin_fma(t, g_dp.pinv, DP_C1);
        const double q = __builtin_fma(t, g_dp.pinv, DP_C2 - hi) - 0x1p52;
#pragma unroll
        for (int j = 0; j < 5; ++j) mul_acc(q, g_dp.p[j], c[i + j + 1], c[i + j]);
        c[i + 1] += c[i] >> 52;     // column i is complete and a multiple of 2^52
    }
    dpe r;
#pragma unroll
    for (int k = 5; k < 10; ++k) {
        r.l[k - 5] = to_double52(c[k] & MASK52);
        c[k + 1] += c[k] >> 52;
    }
    return r;
}

__global__ void k_dp_check(const double* a, const double* b, double* out, size_t n) {
    set_round_toward_zero_f64();
    size_t gid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (gid >= n) return;
    dpe x, y;
    for (int i = 0; i < 5; ++i) { x.l[i] = a[gid * 5 + i]; y.l[i] = b[gid * 5 + i]; }
    dpe z = dp_mul(x, y);
    for (int i = 0; i < 5; ++i) out[gid * 5 + i] = z.l[i];
}
__global__ void k_dp_chain(const double* a, double* out, int iters) {
    set_round_toward_zero_f64();
    size_t gid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    dpe x, y;
    for (int i = 0; i < 5; ++i) { x.l[i] = a[(gid & 1023) * 5 + i]; y.l[i] = a[((gid + 511) & 1023) * 5 + i]; }
    for (int it = 0; it < iters; ++it) { x = dp_mul(x, y); y = dp_mul(y, x); }
    for (int i = 0; i < 5; ++i) out[gid * 5 + i] = x.l[i] + y.l[i];
}
// the product this repo uses today, same dependency shape
__device__ __forceinline__ el2<Fq> seeded(uint64_t seed, uint64_t gid) { return reduce(el<Fq, 32 * U>(fe_split<5>(synth_raw253(seed, gid)))); }
__global__ void k_int_chain(uint32_t* out, int iters, uint64_t seed) {
    uint64_t gid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    el2<Fq> x = seeded(seed, gid), y = seeded(seed + 1, gid);
    for (int it = 0; it < iters; ++it) { x = x * y; y = y * x; }
    store_raw<Fq>(out + gid * 8, x + y);
}

// ---- host side: 256-bit integers as 4 x u64
static void to_limbs52(const uint64_t w[4], double out[5]) {
    for (int i = 0; i < 5; ++i) {
        int bit = 52 * i, q = bit / 64, r = bit % 64;
        uint64_t v = w[q] >> r;
        if (r > 12 && q + 1 < 4) v |= w[q + 1] << (64 - r);
        out[i] = (double)(v & 0xfffffffffffffull);
    }
}
static bool from_limbs52(const double in[5], uint64_t w[5]) {   // -> 320-bit value; false if a limb is not an integer in [0, 2^52)
    for (int i = 0; i < 5; ++i) w[i] = 0;
    for (int i = 0; i < 5; ++i) {
        if (!(in[i] >= 0 && in[i] < 0x1p52) || in[i] != (double)(uint64_t)in[i]) return false;
        uint64_t v = (uint64_t)in[i];
        int bit = 52 * i, q = bit / 64, r = bit % 64;
        w[q] |= v << r;
        if (r > 12) w[q + 1] |= v >> (64 - r);
    }
    return true;
}
static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

template <class F>
static double time_ms(F launch, int reps = 5) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main(int argc, char** argv) {
    const uint64_t* P = hostfq::P;
    DpConst hc;
    to_limbs52(P, hc.p);
    // -p^-1 mod 2^52 by Newton iteration on the low limb
    uint64_t p0 = P[0], inv = 1;
    for (int i = 0; i < 6; ++i) inv *= 2 - p0 * inv;
    hc.pinv = (double)((0 - inv) & 0xfffffffffffffull);
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_dp), &hc, sizeof(hc)));
    // T = 2^260 mod p (for the check z 2^260 == x y): 2^256 mod p doubled four times
    uint64_t T[4];
    for (int i = 0; i < 4; ++i) T[i] = hostfq::ONE[i];
    for (int i = 0; i < 4; ++i) hostmont::add(T, T, T, P);

    size_t n = argc > 1 ? strtoull(argv[1], 0, 0) : (1u << 20);
    std::vector<uint64_t> xs(n * 4), ys(n * 4);
    auto put = [&](std::vector<uint64_t>& v, size_t i, const uint64_t* w) { for (int j = 0; j < 4; ++j) v[i * 4 + j] = w[j]; };
    uint64_t pm1[4] = {P[0] - 1, P[1], P[2], P[3]}, zero[4] = {0, 0, 0, 0}, one[4] = {1, 0, 0, 0};
    uint64_t twop_m1[4];   // 2 p - 1: the largest lazy operand
    { unsigned __int128 c = 0; for (int i = 0; i < 4; ++i) { c += (unsigned __int128)P[i] * 2; twop_m1[i] = (uint64_t)c; c >>= 64; } twop_m1[0] -= 1; }
    uint64_t allones52[4] = {~0ull, ~0ull, ~0ull, 0x3fffffffffffffffull};   // 2^254 - 1: all-ones limbs (still < 2 p)
    const uint64_t* edges[] = {zero, one, pm1, twop_m1, P, allones52};
    size_t ne = 6;
    for (size_t i = 0; i < n; ++i) {
        uint64_t a[4], b[4];
        if (i < ne * ne) { put(xs, i, edges[i / ne]); put(ys, i, edges[i % ne]); continue; }
        for (int j = 0; j < 4; ++j) { a[j] = rnd(); b[j] = rnd(); }
        a[3] &= 0x3fffffffffffffffull; b[3] &= 0x3fffffffffffffffull;      // < 2^254
        while (hostmont::geq(a, P)) hostmont::sub_mod(a, P);
        while (hostmont::geq(b, P)) hostmont::sub_mod(b, P);
        if (i & 1) { unsigned __int128 c = 0; for (int j = 0; j < 4; ++j) { c += (unsigned __int128)a[j] + P[j]; a[j] = (uint64_t)c; c >>= 64; } }   // lazy operands in [p, 2 p)
        put(xs, i, a); put(ys, i, b);
    }
    std::vector<double> ha(n * 5), hb(n * 5), hz(n * 5);
    for (size_t i = 0; i < n; ++i) { to_limbs52(&xs[i * 4], &ha[i * 5]); to_limbs52(&ys[i * 4], &hb[i * 5]); }
    double *da, *db, *dz;
    CK(hipMalloc(&da, n * 40)); CK(hipMalloc(&db, n * 40)); CK(hipMalloc(&dz, n * 40));
    CK(hipMemcpy(da, ha.data(), n * 40, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), n * 40, hipMemcpyHostToDevice));
    k_dp_check<<<(unsigned)((n + 255) / 256), 256>>>(da, db, dz, n);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hz.data(), dz, n * 40, hipMemcpyDeviceToHost));
    size_t bad = 0, over = 0;
    for (size_t i = 0; i < n; ++i) {
        uint64_t z[5];
        if (!from_limbs52(&hz[i * 5], z)) { if (bad++ < 5) printf("  element %zu: a limb is not a 52-bit integer\n", i); continue; }
        // z < 2 p expected for operands < 2 p; reduce fully for the comparison (z[4] holds bits 256..)
        uint64_t zr[4] = {z[0], z[1], z[2], z[3]};
        bool in_bound = z[4] == 0 && !hostmont::geq(zr, P) ? true : (z[4] == 0);
        if (z[4] != 0) { if (i >= ne * ne) ++over; }
        (void)in_bound;
        // value mod p: fold z[4] 2^256 = z[4] * ONE
        uint64_t acc[4] = {zr[0], zr[1], zr[2], zr[3]};
        while (hostmont::geq(acc, P)) hostmont::sub_mod(acc, P);
        for (uint64_t t = 0; t < z[4]; ++t) hostmont::add(acc, acc, hostfq::ONE, P);
        uint64_t x[4], y[4], w1[4], w2[4];
        for (int j = 0; j < 4; ++j) { x[j] = xs[i * 4 + j]; y[j] = ys[i * 4 + j]; }
        while (hostmont::geq(x, P)) hostmont::sub_mod(x, P);
        while (hostmont::geq(y, P)) hostmont::sub_mod(y, P);
        hostmont::mul(w1, x, y, P, hostfq::INV);        // x y / 2^256
        hostmont::mul(w2, acc, T, P, hostfq::INV);      // z 2^260 / 2^256
        bool ok = true;
        for (int j = 0; j < 4; ++j) ok &= w1[j] == w2[j];
        if (!ok && bad++ < 5) printf("  element %zu: z 2^260 != x y (mod p)\n", i);
    }
    printf("exactness: %zu products (36 edge pairs + random canonical / lazy operands), %zu wrong, %zu random results >= 2^256\n", n, bad, over);
    if (bad) return 2;

    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    printf("device %s CUs %d\n", prop.name, cus);
    uint32_t* out; CK(hipMalloc(&out, 64ull << 20));
    const int iters = 500;
    for (int wpc : {4, 8, 16, 32}) {   // waves per CU = 1 / 2 / 4 / 8 per SIMD
        int blocks = cus * wpc / 4, threads = 256;
        double lanes = (double)blocks * threads;
        double ms_i = time_ms([&] { k_int_chain<<<blocks, threads>>>(out, iters, 1); });
        double ms_d = time_ms([&] { k_dp_chain<<<blocks, threads>>>(da, (double*)out, iters); });
        printf("waves/SIMD=%d: 9x29-bit v_mad_u64_u32 product %.3f ms = %.2f G/s | 5x52-bit v_fma_f64 product %.3f ms = %.2f G/s | ratio fp64/int = %.3f\n",
               wpc / 4, ms_i, lanes * 2 * iters / ms_i / 1e6, ms_d, lanes * 2 * iters / ms_d / 1e6, ms_i / ms_d);
    }
    return 0;
}
