// bn254.hpp — BN254 Fq / Fr arithmetic and the G1 group law for gfx950 (host + device).
//
// Replaces halo2curves 0.4.0 src/bn256/{fq,fr,curve}.rs + src/derive/{field,curve}.rs on the device
// [UPSTREAM-RECALL; crate pinned at /root/reference/Cargo.lock:1359-1361].
//
// Representation (chosen for the CDNA4 integer pipe, measured in profiles/r01_microbench_gfx950.txt:
// v_mad_u64_u32 is the only wide multiplier, ~5 cycles per wave-instruction, and has no carry-in):
//   * in registers a field element is 9 limbs of 29 bits ("fe"), value possibly unreduced
//     (< B*p for a compile-time bound B <= 120, 128 p < 2^261).  29-bit limbs leave 6 spare bits in
//     a 64-bit column accumulator, so a Montgomery product is 171 chained v_mad_u64_u32 with NO
//     carry handling until one final pass, against ~600 instructions for the 8x32-bit CIOS form.
//   * Montgomery radix is R' = 2^261 (nine 29-bit reduction steps).  mul(a, b) = a*b/R' mod p.
//   * in memory (and across the C ABI) an element is 8 x u32 = halo2curves' 4 x u64 Montgomery form
//     with R = 2^256, canonical (< p).  Conversions:
//       load_raw   : v            (B = 1)   — for linear pipelines (NTT: scale-preserving) and for
//                                              tables this library stores itself in R' form
//       load_x32   : 32 * v       (B = 32)  — exactly the R' form of the same field element, unreduced
//       store_raw  : canonical(a)
//       store_div32: canonical(a / 32)      — back from the R' form to the ABI form
//   * bounds are carried in the type (el<P, B>) and checked by static_assert: a subtraction adds
//     (B_b + 1) p so it never goes negative; a product contracts to (A*B/128 + 1) p.
//
// No MFMA: nothing here is a dense contraction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ZK_HD __host__ __device__

namespace zk {

constexpr uint32_t LB = 29;
constexpr uint32_t LMASK = (1u << LB) - 1;

struct fe {
    uint32_t l[9];
};
struct alignas(16) fe32 {  // memory / ABI form
    uint32_t w[8];
};

struct FqP {
    static constexpr uint32_t M[9] = {0x187cfd47u, 0x010460b6u, 0x1c72a34fu, 0x02d522d0u, 0x1585d978u,
                                      0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
    static constexpr uint32_t INV = 0x04866389u;  // -p^-1 mod 2^29
    static constexpr uint32_t ONE[9] = {0x157ccc21u, 0x141c2758u, 0x185230d3u, 0x014c0419u, 0x0aa36fb9u,
                                        0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};  // 2^261 mod p
    static constexpr uint32_t R2[9] = {0x059bac10u, 0x0d1503a3u, 0x018016b8u, 0x10ab0ca8u, 0x02632639u,
                                       0x02c0169fu, 0x169bfd53u, 0x11869d4cu, 0x002a11a6u};   // 2^522 mod p
    static constexpr uint32_t NEGINV32 = 9;       // -p^-1 mod 32
};
struct FrP {
    static constexpr uint32_t M[9] = {0x10000001u, 0x1f0fac9fu, 0x0e5c2450u, 0x07d090f3u, 0x1585d283u,
                                      0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
    static constexpr uint32_t INV = 0x0fffffffu;
    static constexpr uint32_t ONE[9] = {0x0fffff57u, 0x1ea70ab4u, 0x052c068bu, 0x17504f49u, 0x0aa8075bu,
                                        0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};
    static constexpr uint32_t R2[9] = {0x05b69bd4u, 0x06170a5au, 0x020cddceu, 0x1db6310bu, 0x0e54d0ffu,
                                       0x1cf855e3u, 0x1c15e103u, 0x07d09161u, 0x000a054au};
    static constexpr uint32_t NEGINV32 = 31;
};
using Fq = FqP;
using Fr = FrP;
constexpr uint64_t P8_MAGIC34 = 5417ull;  // floor(2^34 / ((p >> 232) + 1)), same for both moduli

// limb i of k*p, normalised (compile-time for constant k); the top limb keeps everything above bit 232.
template <class P>
ZK_HD constexpr uint32_t kp_limb(uint32_t k, int i) {
    uint64_t c = 0;
    uint32_t r = 0;
    for (int j = 0; j <= i; ++j) {
        c += (uint64_t)k * P::M[j];
        r = (j < 8) ? (uint32_t)(c & LMASK) : (uint32_t)c;
        c >>= LB;
    }
    return r;
}
// limb i of k*p in "borrow-spread" form: every limb below the top is >= 2^29 - 1, so that
// spread(k p) - b is limb-wise non-negative for any normalised b with top limb <= top(k p) - 1.
template <class P>
ZK_HD constexpr uint32_t kp_spread(uint32_t k, int i) {
    return i == 0 ? kp_limb<P>(k, 0) + (1u << LB) : i < 8 ? kp_limb<P>(k, i) + LMASK : kp_limb<P>(k, 8) - 1;
}

// ---------------------------------------------------------------------------------- raw limb ops
ZK_HD __forceinline__ void fe_normalize(fe& a) {  // limbs < 2^32 in, limbs < 2^29 (top: whatever is left) out
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a.l[i + 1] += a.l[i] >> LB;
        a.l[i] &= LMASK;
    }
}
ZK_HD __forceinline__ fe fe_zero() {
    fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = 0;
    return r;
}
ZK_HD __forceinline__ bool fe_is_zero_exact(const fe& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) o |= a.l[i];
    return o == 0;
}

// a * b / 2^261 mod p; normalised inputs of any bounds A, B; output < (A*B/128 + 1) p, normalised.
template <class P>
ZK_HD __forceinline__ fe fe_mul_raw(const fe& a, const fe& b) {
    uint64_t acc[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) acc[j] = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
#pragma unroll
        for (int j = 0; j < 9; ++j) acc[j] += (uint64_t)a.l[j] * b.l[i];
        uint32_t q = ((uint32_t)acc[0] * P::INV) & LMASK;
#pragma unroll
        for (int j = 0; j < 9; ++j) acc[j] += (uint64_t)q * P::M[j];
        uint64_t carry = acc[0] >> LB;  // low 29 bits are zero by construction
        acc[0] = acc[1] + carry;
#pragma unroll
        for (int j = 1; j < 8; ++j) acc[j] = acc[j + 1];
        acc[8] = 0;
    }
    fe r;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        r.l[j] = (uint32_t)acc[j] & LMASK;
        acc[j + 1] += acc[j] >> LB;
    }
    r.l[8] = (uint32_t)acc[8];
    return r;
}

// (a * b + c * d) / 2^261 mod p with ONE reduction: 243 multiplier instructions instead of 324.  Normalised inputs of bounds A, B, C, D;
// output < ((A*B + C*D)/128 + 1) p.  Columns: 9 iterations x (2 + 1) products < 2^58 each stay below 2^63.
template <class P>
ZK_HD __forceinline__ fe fe_mul2_raw(const fe& a, const fe& b, const fe& c, const fe& d) {
    uint64_t acc[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) acc[j] = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
#pragma unroll
        for (int j = 0; j < 9; ++j) acc[j] += (uint64_t)a.l[j] * b.l[i];
#pragma unroll
        for (int j = 0; j < 9; ++j) acc[j] += (uint64_t)c.l[j] * d.l[i];
        uint32_t q = ((uint32_t)acc[0] * P::INV) & LMASK;
#pragma unroll
        for (int j = 0; j < 9; ++j) acc[j] += (uint64_t)q * P::M[j];
        uint64_t carry = acc[0] >> LB;
        acc[0] = acc[1] + carry;
#pragma unroll
        for (int j = 1; j < 8; ++j) acc[j] = acc[j + 1];
        acc[8] = 0;
    }
    fe r;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        r.l[j] = (uint32_t)acc[j] & LMASK;
        acc[j + 1] += acc[j] >> LB;
    }
    r.l[8] = (uint32_t)acc[8];
    return r;
}

// sum_{k < N} a_k * b_k / 2^261 mod p with ONE reduction (N <= 4: columns stay below (N + 1) * 9 * 2^58 < 2^64).
// Normalised inputs; output < (sum A_k B_k / 128 + 1) p.
template <class P, int N>
ZK_HD __forceinline__ fe fe_dot_raw(const fe* a, const fe* b) {
    static_assert(N >= 1 && N <= 4, "fe_dot_raw: at most four products per reduction");
    uint64_t acc[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) acc[j] = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
#pragma unroll
        for (int k = 0; k < N; ++k)
#pragma unroll
            for (int j = 0; j < 9; ++j) acc[j] += (uint64_t)a[k].l[j] * b[k].l[i];
        uint32_t q = ((uint32_t)acc[0] * P::INV) & LMASK;
#pragma unroll
        for (int j = 0; j < 9; ++j) acc[j] += (uint64_t)q * P::M[j];
        uint64_t carry = acc[0] >> LB;
        acc[0] = acc[1] + carry;
#pragma unroll
        for (int j = 1; j < 8; ++j) acc[j] = acc[j + 1];
        acc[8] = 0;
    }
    fe r;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        r.l[j] = (uint32_t)acc[j] & LMASK;
        acc[j + 1] += acc[j] >> LB;
    }
    r.l[8] = (uint32_t)acc[8];
    return r;
}

// a * a / 2^261 mod p: the 36 cross products once (doubled), 9 squares, then the 81 reduction products:
// 126 multiplier instructions instead of 171.  Columns stay below 2 * 4 * 2^58 + 2^58 + 9 * 2^58 < 2^63.
template <class P>
ZK_HD __forceinline__ fe fe_sqr_raw(const fe& a) {
    uint64_t col[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) col[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int j = i + 1; j < 9; ++j) col[i + j] += (uint64_t)a.l[i] * a.l[j];
#pragma unroll
    for (int k = 1; k < 16; ++k) col[k] <<= 1;
#pragma unroll
    for (int i = 0; i < 9; ++i) col[2 * i] += (uint64_t)a.l[i] * a.l[i];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        uint32_t q = ((uint32_t)col[i] * P::INV) & LMASK;
#pragma unroll
        for (int j = 0; j < 9; ++j) col[i + j] += (uint64_t)q * P::M[j];
        col[i + 1] += col[i] >> LB;   // low 29 bits of col[i] are zero now
    }
    fe r;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        r.l[j] = (uint32_t)col[9 + j] & LMASK;
        col[10 + j] += col[9 + j] >> LB;
    }
    r.l[8] = (uint32_t)col[17];
    return r;
}

// returns a - k p if that is >= 0, else a  (runtime k <= 127)
template <class P>
ZK_HD __forceinline__ fe fe_cond_sub_kp(const fe& a, uint32_t k) {
    fe d;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        c += (int64_t)a.l[i] - (int64_t)((uint64_t)k * P::M[i]);
        d.l[i] = i < 8 ? (uint32_t)c & LMASK : (uint32_t)c;
        c >>= LB;  // arithmetic shift
    }
    bool neg = c < 0;
    fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = neg ? a.l[i] : d.l[i];
    return r;
}
// canonical representative (< p) of any normalised a < 126 p.  Used at stores and zero tests only.
template <class P>
ZK_HD __forceinline__ fe fe_canonical(const fe& a_in) {
    fe a = a_in;
    uint32_t k = (uint32_t)(((uint64_t)a.l[8] * P8_MAGIC34) >> 34);  // underestimates floor(a / p) by at most 3
    a = fe_cond_sub_kp<P>(a, k);
    a = fe_cond_sub_kp<P>(a, 2);
    a = fe_cond_sub_kp<P>(a, 1);
    return a;
}
template <class P>
ZK_HD __forceinline__ bool fe_is_zero_modp(const fe& a) {
    return fe_is_zero_exact(fe_canonical<P>(a));
}

// 8 x u32 <-> 9 x 29-bit.  SHIFT = 0: value v;  SHIFT = 5: value 32 v.
template <int SHIFT>
ZK_HD __forceinline__ fe fe_split(const fe32& m) {
    fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        int bit = 29 * i - SHIFT;  // first source bit of limb i
        uint32_t v;
        if (bit < 0) {
            v = m.w[0] << (-bit);
        } else {
            int wi = bit >> 5, sh = bit & 31;
            uint32_t lo = wi < 8 ? m.w[wi] : 0u, hi = wi + 1 < 8 ? m.w[wi + 1] : 0u;
            v = sh ? (lo >> sh) | (hi << (32 - sh)) : lo;
        }
        r.l[i] = i < 8 ? (v & LMASK) : v;
    }
    return r;
}
ZK_HD __forceinline__ fe32 fe_pack(const fe& a) {  // canonical limbs (value < 2^256)
    fe32 m;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        int bit = 32 * j, li = bit / 29, sh = bit % 29;
        uint64_t v = (uint64_t)a.l[li] >> sh;
        int have = 29 - sh;
        if (li + 1 < 9) v |= (uint64_t)a.l[li + 1] << have;
        if (have + 29 < 32 && li + 2 < 9) v |= (uint64_t)a.l[li + 2] << (have + 29);
        m.w[j] = (uint32_t)v;
    }
    return m;
}
ZK_HD __forceinline__ fe32 mem_load(const void* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1];
    fe32 m;
    m.w[0] = a.x; m.w[1] = a.y; m.w[2] = a.z; m.w[3] = a.w;
    m.w[4] = b.x; m.w[5] = b.y; m.w[6] = b.z; m.w[7] = b.w;
    return m;
}
ZK_HD __forceinline__ void mem_store(void* p, const fe32& m) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(m.w[0], m.w[1], m.w[2], m.w[3]);
    q[1] = make_uint4(m.w[4], m.w[5], m.w[6], m.w[7]);
}
// exact division by 32 mod p: add k p with k = -a p^-1 mod 32, then shift.  a < 88 p.
template <class P>
ZK_HD __forceinline__ fe fe_div32(const fe& a) {
    uint32_t k = (a.l[0] * P::NEGINV32) & 31u;
    uint64_t c = 0;
    uint32_t t[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        c += (uint64_t)a.l[i] + (uint64_t)k * P::M[i];
        t[i] = i < 8 ? (uint32_t)c & LMASK : (uint32_t)c;
        c >>= LB;
    }
    fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        uint32_t lo = t[i] >> 5, hi = i + 1 < 9 ? (t[i + 1] & 31u) << (LB - 5) : 0u;
        r.l[i] = i < 8 ? (lo | hi) & LMASK : lo;
    }
    return r;
}

// ---------------------------------------------------------------------------------- typed elements
// el<P, B>: value < (B / 16) p, limbs normalised.  Bounds are in sixteenths of p so that the +1 p of
// every Montgomery contraction does not compound into uselessly loose integer bounds.
constexpr int U = 16;                 // one p
constexpr int BMAX = 120 * U;         // 120 p < 2^261 with room for fe_canonical's estimate
constexpr int mul_bound(int a, int b) { return (a * b + 128 * U - 1) / (128 * U) + U; }
constexpr int ceil_p(int b) { return (b + U - 1) / U; }

template <class P, int B>
struct el {
    static_assert(B >= 1 && B <= BMAX, "lazy bound out of range (value must stay below 2^261 and canonicalise)");
    fe v;
    ZK_HD el() {}
    ZK_HD explicit el(const fe& f) : v(f) {}
    template <int A>
    ZK_HD el(const el<P, A>& o) : v(o.v) { static_assert(A <= B, "narrowing a lazy bound"); }
};

template <class P, int A, int B>
ZK_HD __forceinline__ el<P, mul_bound(A, B)> operator*(const el<P, A>& a, const el<P, B>& b) {
    return el<P, mul_bound(A, B)>(fe_mul_raw<P>(a.v, b.v));
}
// a0 b0 + a1 b1 + a2 b2 + a3 b3 with one reduction (all eight operands < 2p)
template <class P>
ZK_HD __forceinline__ el<P, mul_bound(4 * U, 4 * U) + 0> dot4(const el<P, 2 * U>& a0, const el<P, 2 * U>& b0, const el<P, 2 * U>& a1, const el<P, 2 * U>& b1,
                                                            const el<P, 2 * U>& a2, const el<P, 2 * U>& b2, const el<P, 2 * U>& a3, const el<P, 2 * U>& b3) {
    // 4 * (2p * 2p) / 128 + 1 p = (16 p^2 / 128 p) + p < 2p: the bound of a product of two 4p values
    fe a[4] = {a0.v, a1.v, a2.v, a3.v}, b[4] = {b0.v, b1.v, b2.v, b3.v};
    return el<P, mul_bound(4 * U, 4 * U)>(fe_dot_raw<P, 4>(a, b));
}
// a * b + c * d with one reduction
constexpr int mul2_bound(int a, int b, int c, int d) { return (a * b + c * d + 128 * U - 1) / (128 * U) + U; }
template <class P, int A, int B, int C, int D>
ZK_HD __forceinline__ el<P, mul2_bound(A, B, C, D)> muladd2(const el<P, A>& a, const el<P, B>& b, const el<P, C>& c, const el<P, D>& d) {
    return el<P, mul2_bound(A, B, C, D)>(fe_mul2_raw<P>(a.v, b.v, c.v, d.v));
}
// A difference / negation that is ONLY a multiplicand of the two-product accumulation needs no carry pass: ell<P, B> is a value
// < (B / 16) p whose limbs are NOT normalised (each < 3 * 2^29: a normalised limb < 2^29 plus a borrow-spread constant <= 2^30).
// Columns of fe_mul2_raw with a normalised x lazy and a lazy x normalised product: each of the 9 steps of a column adds at most
// 2^29 * 3 * 2^29 (a x lazy b) + 3 * 2^29 * 2^29 (lazy c x d) + 2^29 * 2^29 (m x p) = 7 * 2^58, so a column stays below 63 * 2^58 =
// 0.984 * 2^64 — under 2 % of headroom, which is why the limb bound of the spread constant is a static_assert and not a comment.
// Not accepted by anything else (a squaring's or a subtraction's operand must be normalised).
template <class P>
ZK_HD constexpr bool kp_spread_fits(uint32_t k) {   // every limb of spread(k p) <= 2^30 (top limb: < 2^29, it carries no borrow constant)
    for (int i = 0; i < 8; ++i)
        if (kp_spread<P>(k, i) > (1u << 30)) return false;
    return kp_spread<P>(k, 8) < (1u << LB);
}
template <class P, int B>
struct ell {
    static_assert(B >= 1 && B <= BMAX, "lazy bound out of range (value must stay below 2^261)");
    fe v;
};
template <class P, int A, int B>
ZK_HD __forceinline__ ell<P, A + (ceil_p(B) + 1) * U> sub_lazy(const el<P, A>& a, const el<P, B>& b) {
    static_assert(kp_spread_fits<P>(ceil_p(B) + 1), "a limb of spread(k p) exceeds 2^30: the 63 * 2^58 column bound of the lazy muladd2 no longer holds");
    ell<P, A + (ceil_p(B) + 1) * U> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.v.l[i] = a.v.l[i] + kp_spread<P>(ceil_p(B) + 1, i) - b.v.l[i];
    return r;
}
template <class P, int B>
ZK_HD __forceinline__ ell<P, (ceil_p(B) + 1) * U> neg_lazy(const el<P, B>& b) {
    static_assert(kp_spread_fits<P>(ceil_p(B) + 1), "a limb of spread(k p) exceeds 2^30: the 63 * 2^58 column bound of the lazy muladd2 no longer holds");
    ell<P, (ceil_p(B) + 1) * U> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.v.l[i] = kp_spread<P>(ceil_p(B) + 1, i) - b.v.l[i];
    return r;
}
template <class P, int A, int B, int C, int D>
ZK_HD __forceinline__ el<P, mul2_bound(A, B, C, D)> muladd2(const el<P, A>& a, const ell<P, B>& b, const ell<P, C>& c, const el<P, D>& d) {
    static_assert(mul2_bound(A, B, C, D) <= BMAX, "lazy bound out of range");
    return el<P, mul2_bound(A, B, C, D)>(fe_mul2_raw<P>(a.v, b.v, c.v, d.v));
}
template <class P, int A>
ZK_HD __forceinline__ el<P, mul_bound(A, A)> sqr(const el<P, A>& a) {
    return el<P, mul_bound(A, A)>(fe_sqr_raw<P>(a.v));
}
template <class P, int A, int B>
ZK_HD __forceinline__ el<P, A + B> operator+(const el<P, A>& a, const el<P, B>& b) {
    fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = a.v.l[i] + b.v.l[i];
    fe_normalize(r);
    return el<P, A + B>(r);
}
// a - b + (ceil(B) + 1) p
template <class P, int A, int B>
ZK_HD __forceinline__ el<P, A + (ceil_p(B) + 1) * U> operator-(const el<P, A>& a, const el<P, B>& b) {
    fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = a.v.l[i] + kp_spread<P>(ceil_p(B) + 1, i) - b.v.l[i];
    fe_normalize(r);
    return el<P, A + (ceil_p(B) + 1) * U>(r);
}
template <class P, int B>
ZK_HD __forceinline__ el<P, (ceil_p(B) + 1) * U> neg(const el<P, B>& b) {
    fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = kp_spread<P>(ceil_p(B) + 1, i) - b.v.l[i];
    fe_normalize(r);
    return el<P, (ceil_p(B) + 1) * U>(r);
}
// a * K for a small constant K (K * 2^29 must fit 32 bits: K <= 8)
template <int K, class P, int A>
ZK_HD __forceinline__ el<P, K * A> mul_small(const el<P, A>& a) {
    static_assert(K >= 1 && K <= 8, "small multiple");
    fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = a.v.l[i] * (uint32_t)K;
    fe_normalize(r);
    return el<P, K * A>(r);
}
// a + K b with ONE carry pass (K <= 4: limbs stay below (1 + K) 2^29 < 2^32)
template <int K, class P, int A, int B>
ZK_HD __forceinline__ el<P, A + K * B> add_mul_small(const el<P, A>& a, const el<P, B>& b) {
    static_assert(K >= 1 && K <= 4, "small multiple");
    fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = a.v.l[i] + b.v.l[i] * (uint32_t)K;
    fe_normalize(r);
    return el<P, A + K * B>(r);
}
// a == 0 mod p?  If a = k p with k < A/16 then k = a_0 * p_0^-1 mod 2^29: three instructions reject every
// other value; the exact test runs only when the filter fires (probability ~ k_max / 2^29 per lane).
template <class P, int A>
ZK_HD __forceinline__ bool is_zero(const el<P, A>& a) {
    constexpr uint32_t PINV = ((1u << LB) - P::INV) & LMASK;   // p^-1 mod 2^29
    uint32_t k = (a.v.l[0] * PINV) & LMASK;
    if (k >= (uint32_t)ceil_p(A)) return false;
    return fe_is_zero_modp<P>(a.v);
}
// the filter of is_zero alone: false means a != 0 mod p for sure (three instructions); true needs the exact test
template <class P, int A>
ZK_HD __forceinline__ bool maybe_zero(const el<P, A>& a) {
    constexpr uint32_t PINV = ((1u << LB) - P::INV) & LMASK;
    return ((a.v.l[0] * PINV) & LMASK) < (uint32_t)ceil_p(A);
}
template <class P, int A, int B>
ZK_HD __forceinline__ bool equal(const el<P, A>& a, const el<P, B>& b) { return is_zero(a - b); }
template <class P, int A>
ZK_HD __forceinline__ el<P, U> canonical(const el<P, A>& a) { return el<P, U>(fe_canonical<P>(a.v)); }
// re-contract a grown value: a * 1 (in R' form) < (A/128 + 1) p
template <class P, int A>
ZK_HD __forceinline__ el<P, mul_bound(A, U)> reduce(const el<P, A>& a) {
    fe o;
#pragma unroll
    for (int i = 0; i < 9; ++i) o.l[i] = P::ONE[i];
    return el<P, mul_bound(A, U)>(fe_mul_raw<P>(a.v, o));
}
template <class P, int A>
ZK_HD __forceinline__ el<P, A> select(bool c, const el<P, A>& a, const el<P, A>& b) {  // c ? a : b
    el<P, A> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.v.l[i] = c ? a.v.l[i] : b.v.l[i];
    return r;
}

template <class P>
ZK_HD __forceinline__ el<P, U> one() {
    fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = P::ONE[i];
    return el<P, U>(r);
}
template <class P>
ZK_HD __forceinline__ el<P, U> zero() { return el<P, U>(fe_zero()); }
template <class P>
using el1 = el<P, U>;        // canonical-sized (< p)
template <class P>
using el2 = el<P, 2 * U>;    // product-sized (< 2p)

// memory <-> registers (see the header comment for which one to use where)
template <class P>
ZK_HD __forceinline__ el<P, U> load_raw(const void* p) { return el<P, U>(fe_split<0>(mem_load(p))); }
template <class P>
ZK_HD __forceinline__ el<P, 32 * U> load_x32(const void* p) { return el<P, 32 * U>(fe_split<5>(mem_load(p))); }
template <class P, int A>
ZK_HD __forceinline__ void store_raw(void* p, const el<P, A>& a) { mem_store(p, fe_pack(fe_canonical<P>(a.v))); }
// a value < 4p (< 2^256) packed as it is, NOT reduced below p: for intermediate buffers whose readers take lazy inputs (the NTT's
// pass-to-pass buffers: the next pass loads "< 2p" operands anyway) — saves fe_canonical's three conditional subtractions
template <class P, int A>
ZK_HD __forceinline__ void store_packed(void* p, const el<P, A>& a) {
    static_assert(A <= 4 * U, "store_packed: the value must fit 256 bits");
    mem_store(p, fe_pack(a.v));
}
template <class P, int A>
ZK_HD __forceinline__ fe32 to_abi(const el<P, A>& a) {
    static_assert(A <= 88 * U, "to_abi input bound");
    return fe_pack(fe_canonical<P>(fe_div32<P>(a.v)));
}
template <class P, int A>
ZK_HD __forceinline__ void store_div32(void* p, const el<P, A>& a) { mem_store(p, to_abi(a)); }
// ABI value (x 2^256, canonical) -> exact R' form (x 2^261), reduced (< 2p): 32 v, then one product by "one"
template <class P>
ZK_HD __forceinline__ el2<P> from_abi(const fe32& m) { return reduce(el<P, 32 * U>(fe_split<5>(m))); }
// plain integer x < 2^64 -> R' form
template <class P>
ZK_HD __forceinline__ el2<P> from_u64(uint64_t x) {
    fe t = fe_zero(), r2;
    t.l[0] = (uint32_t)(x & LMASK);
    t.l[1] = (uint32_t)((x >> 29) & LMASK);
    t.l[2] = (uint32_t)(x >> 58);
#pragma unroll
    for (int i = 0; i < 9; ++i) r2.l[i] = P::R2[i];
    return el2<P>(fe_mul_raw<P>(t, r2));
}
// canonical 8 x u32 integer (not Montgomery) -> R' form
template <class P>
ZK_HD __forceinline__ el2<P> from_canonical_words(const uint32_t w[8]) {
    fe32 m;
#pragma unroll
    for (int i = 0; i < 8; ++i) m.w[i] = w[i];
    fe r2;
#pragma unroll
    for (int i = 0; i < 9; ++i) r2.l[i] = P::R2[i];
    return el2<P>(fe_mul_raw<P>(fe_split<0>(m), r2));
}
// R' form -> canonical integer words (the field element itself)
template <class P, int A>
ZK_HD __forceinline__ fe32 to_canonical_words(const el<P, A>& a) {
    fe o = fe_zero();
    o.l[0] = 1;
    return fe_pack(fe_canonical<P>(fe_mul_raw<P>(a.v, o)));
}
// ABI words (x 2^256 canonical) -> canonical integer words of x: (v * 32) / 2^261
template <class P>
ZK_HD __forceinline__ fe32 abi_to_canonical_words(const fe32& m) {
    fe o = fe_zero();
    o.l[0] = 32;
    return fe_pack(fe_canonical<P>(fe_mul_raw<P>(fe_split<0>(m), o)));
}

// a^e; results < 2 p
template <class P>
ZK_HD inline el2<P> pow_u64(const el2<P>& a, uint64_t e) {
    el2<P> acc = one<P>();
    bool started = false;
    for (int i = 63; i >= 0; --i) {
        if (started) acc = sqr(acc);
        if ((e >> i) & 1) {
            if (started) acc = acc * a; else acc = a;
            started = true;
        }
    }
    return acc;
}
template <class P>
ZK_HD inline el2<P> pow_words(const el2<P>& a, const uint32_t e[8]) {
    el2<P> acc = one<P>();
    bool started = false;
    for (int i = 255; i >= 0; --i) {
        if (started) acc = sqr(acc);
        if ((e[i >> 5] >> (i & 31)) & 1) {
            if (started) acc = acc * a; else acc = a;
            started = true;
        }
    }
    return acc;
}
// Fermat inverse (0 -> 0)
template <class P>
ZK_HD inline el2<P> inv(const el2<P>& a) {
    fe t;
    for (int i = 0; i < 9; ++i) t.l[i] = P::M[i];
    fe32 pm = fe_pack(t);
    uint32_t e[8];
    for (int i = 0; i < 8; ++i) e[i] = pm.w[i];
    e[0] -= 2;
    return pow_words<P>(a, e);
}

// Inverse by the binary extended Euclidean algorithm on 8 x u32 words, for device code that needs ONE inverse on a latency-bound
// path (the root of a batch-inversion tree): ~30 k simple instructions against the Fermat chain's 380 products (~95 k).
// Data-dependent trip counts: call it with the same value in every lane of a wave.  0 -> 0.
template <class P>
ZK_HD inline el2<P> inv_euclid(const el2<P>& a) {
    struct W8 { uint32_t w[8]; };
    auto is_one = [](const W8& x) { uint32_t o = x.w[0] ^ 1u; for (int i = 1; i < 8; ++i) o |= x.w[i]; return o == 0; };
    auto geq = [](const W8& x, const W8& y) {
        for (int i = 7; i >= 0; --i) if (x.w[i] != y.w[i]) return x.w[i] > y.w[i];
        return true;
    };
    auto add = [](W8& x, const W8& y) { uint64_t c = 0; for (int i = 0; i < 8; ++i) { c += (uint64_t)x.w[i] + y.w[i]; x.w[i] = (uint32_t)c; c >>= 32; } };
    auto sub = [](W8& x, const W8& y) { uint64_t b = 0; for (int i = 0; i < 8; ++i) { uint64_t d = (uint64_t)x.w[i] - y.w[i] - b; x.w[i] = (uint32_t)d; b = (d >> 32) & 1; } };
    auto shr1 = [](W8& x) { for (int i = 0; i < 7; ++i) x.w[i] = (x.w[i] >> 1) | (x.w[i + 1] << 31); x.w[7] >>= 1; };
    fe pm;
    for (int i = 0; i < 9; ++i) pm.l[i] = P::M[i];
    W8 p, u, v, b, c;
    {
        fe32 t = fe_pack(pm), q = fe_pack(fe_canonical<P>(a.v));
        uint32_t any = 0;
        for (int i = 0; i < 8; ++i) { p.w[i] = t.w[i]; u.w[i] = q.w[i]; v.w[i] = t.w[i]; b.w[i] = 0; c.w[i] = 0; any |= q.w[i]; }
        if (!any) return el2<P>(fe_zero());
        b.w[0] = 1;
    }
    while (!is_one(u) && !is_one(v)) {
        while (!(u.w[0] & 1)) { shr1(u); if (b.w[0] & 1) add(b, p); shr1(b); }
        while (!(v.w[0] & 1)) { shr1(v); if (c.w[0] & 1) add(c, p); shr1(c); }
        if (geq(u, v)) { sub(u, v); if (!geq(b, c)) add(b, p); sub(b, c); }
        else { sub(v, u); if (!geq(c, b)) add(c, p); sub(c, b); }
    }
    const W8& y = is_one(u) ? b : c;     // (a R')^-1 as an integer: a^-1 R' = y R'^2 = mont(mont(y, R'^2), R'^2)
    fe32 m;
    for (int i = 0; i < 8; ++i) m.w[i] = y.w[i];
    fe r2;
    for (int i = 0; i < 9; ++i) r2.l[i] = P::R2[i];
    return el2<P>(fe_mul_raw<P>(fe_mul_raw<P>(fe_split<0>(m), r2), r2));
}

// Host-side inverse by the binary extended Euclidean algorithm (~1.5 us against ~40 us for the Fermat chain on one core):
// the Fiat-Shamir round trips normalise a batch of commitments on the host, with the GPU waiting.  0 -> 0.
template <class P>
inline el2<P> inv_host(const el2<P>& a) {
    typedef unsigned __int128 u128;
    struct U256 { uint64_t w[4]; };
    auto from_words = [](const fe32& m) { U256 r; for (int i = 0; i < 4; ++i) r.w[i] = (uint64_t)m.w[2 * i] | ((uint64_t)m.w[2 * i + 1] << 32); return r; };
    auto is_one = [](const U256& x) { return x.w[0] == 1 && !(x.w[1] | x.w[2] | x.w[3]); };
    auto is_zero_ = [](const U256& x) { return !(x.w[0] | x.w[1] | x.w[2] | x.w[3]); };
    auto geq = [](const U256& x, const U256& y) { for (int i = 3; i >= 0; --i) if (x.w[i] != y.w[i]) return x.w[i] > y.w[i]; return true; };
    auto add = [](U256& x, const U256& y) { u128 c = 0; for (int i = 0; i < 4; ++i) { c += (u128)x.w[i] + y.w[i]; x.w[i] = (uint64_t)c; c >>= 64; } };
    auto sub = [](U256& x, const U256& y) { uint64_t b = 0; for (int i = 0; i < 4; ++i) { u128 d = (u128)x.w[i] - y.w[i] - b; x.w[i] = (uint64_t)d; b = (uint64_t)(d >> 64) & 1; } };
    auto shr1 = [](U256& x) { for (int i = 0; i < 3; ++i) x.w[i] = (x.w[i] >> 1) | (x.w[i + 1] << 63); x.w[3] >>= 1; };
    fe pm;
    for (int i = 0; i < 9; ++i) pm.l[i] = P::M[i];
    const U256 p = from_words(fe_pack(pm));
    U256 u = from_words(fe_pack(fe_canonical<P>(a.v))), v = p, b = {{1, 0, 0, 0}}, c = {{0, 0, 0, 0}};
    if (is_zero_(u)) return el2<P>(fe_zero());
    auto halve = [&](U256& x) { if (x.w[0] & 1) add(x, p); shr1(x); };          // x / 2 mod p (x + p < 2^255)
    auto submod = [&](U256& x, const U256& y) { if (!geq(x, y)) add(x, p); sub(x, y); };
    while (!is_one(u) && !is_one(v)) {
        while (!(u.w[0] & 1)) { shr1(u); halve(b); }
        while (!(v.w[0] & 1)) { shr1(v); halve(c); }
        if (geq(u, v)) { sub(u, v); submod(b, c); } else { sub(v, u); submod(c, b); }
    }
    const U256 y = is_one(u) ? b : c;     // (a R')^-1 as an integer: a^-1 R' = y R'^2 = mont(mont(y, R'^2), R'^2)
    fe32 m;
    for (int i = 0; i < 4; ++i) { m.w[2 * i] = (uint32_t)y.w[i]; m.w[2 * i + 1] = (uint32_t)(y.w[i] >> 32); }
    fe r2;
    for (int i = 0; i < 9; ++i) r2.l[i] = P::R2[i];
    return el2<P>(fe_mul_raw<P>(fe_mul_raw<P>(fe_split<0>(m), r2), r2));
}

// Fr constants as canonical integer words (halo2curves src/bn256/fr.rs): 2^28-th root of unity, ZETA, DELTA.
constexpr uint32_t FR_S = 28;
constexpr uint32_t FR_ROOT_OF_UNITY[8] = {0x60c37c9cu, 0xd34f1ed9u, 0xd39329c8u, 0x3215cf6du,
                                          0x3dd31f74u, 0x98865ea9u, 0x166d18b7u, 0x03ddb9f5u};
constexpr uint32_t FR_ZETA[8] = {0xb99c90ddu, 0x8b17ea66u, 0x8d8daaa7u, 0x5bfc4108u, 0x41a91758u, 0xb3c4d79du, 0u, 0u};
constexpr uint32_t FR_DELTA[8] = {0xe533e9a2u, 0x870e56bbu, 0x5e963f25u, 0x5b5f898eu,
                                  0xd4c86e71u, 0x64ec26aau, 0x22c6f0cau, 0x09226b6eu};

// ------------------------------------------------------------------------------------ G1
// y^2 = x^3 + 3.  Register forms (coordinates in R' form):
//   g1a: affine, both coordinates < 2p; identity = exact (0, 0)
//   g1j: Jacobian with the loop invariant X < BX p, Y < BY p, Z < BZ p; identity = exact Z = 0
constexpr int BX = 16 * U, BY = 8 * U, BZ = 4 * U;
struct g1a { el2<Fq> x, y; };
struct g1j { el<Fq, BX> x; el<Fq, BY> y; el<Fq, BZ> z; };

ZK_HD __forceinline__ bool g1a_is_id(const g1a& p) { return fe_is_zero_exact(p.x.v) && fe_is_zero_exact(p.y.v); }
ZK_HD __forceinline__ bool g1j_is_id(const g1j& p) { return fe_is_zero_exact(p.z.v); }
ZK_HD __forceinline__ g1j g1j_identity() {
    g1j r;
    r.x = zero<Fq>(); r.y = one<Fq>(); r.z = zero<Fq>();
    return r;
}
ZK_HD __forceinline__ g1a g1a_identity() {
    g1a r;
    r.x = zero<Fq>(); r.y = zero<Fq>();
    return r;
}
ZK_HD __forceinline__ g1j g1j_from_affine(const g1a& a) {
    if (g1a_is_id(a)) return g1j_identity();
    g1j r;
    r.x = a.x; r.y = a.y; r.z = one<Fq>();
    return r;
}
// table / scratch format: R' form, canonical, 32 B per coordinate
ZK_HD __forceinline__ g1a g1a_load_raw(const void* p) {
    g1a r;
    r.x = load_raw<Fq>(p);
    r.y = load_raw<Fq>(reinterpret_cast<const char*>(p) + 32);
    return r;
}
ZK_HD __forceinline__ void g1a_store_raw(void* p, const g1a& v) {
    store_raw<Fq>(p, v.x);
    store_raw<Fq>(reinterpret_cast<char*>(p) + 32, v.y);
}
ZK_HD __forceinline__ g1j g1j_load_raw(const void* p) {
    g1j r;
    r.x = load_raw<Fq>(p);
    r.y = load_raw<Fq>(reinterpret_cast<const char*>(p) + 32);
    r.z = load_raw<Fq>(reinterpret_cast<const char*>(p) + 64);
    return r;
}
ZK_HD __forceinline__ void g1j_store_raw(void* p, const g1j& v) {
    store_raw<Fq>(p, v.x);
    store_raw<Fq>(reinterpret_cast<char*>(p) + 32, v.y);
    store_raw<Fq>(reinterpret_cast<char*>(p) + 64, v.z);
}
// ABI format (halo2curves G1Affine / G1, R = 2^256)
ZK_HD __forceinline__ g1a g1a_load_abi(const void* p) {
    g1a r;
    r.x = from_abi<Fq>(mem_load(p));
    r.y = from_abi<Fq>(mem_load(reinterpret_cast<const char*>(p) + 32));
    if (fe_is_zero_modp<Fq>(r.x.v) && fe_is_zero_modp<Fq>(r.y.v)) return g1a_identity();
    return r;
}
ZK_HD __forceinline__ void g1a_store_abi(void* p, const g1a& v) {
    mem_store(p, to_abi(v.x));
    mem_store(reinterpret_cast<char*>(p) + 32, to_abi(v.y));
}
ZK_HD __forceinline__ g1j g1j_load_abi(const void* p) {
    g1j r;
    r.x = from_abi<Fq>(mem_load(p));
    r.y = from_abi<Fq>(mem_load(reinterpret_cast<const char*>(p) + 32));
    auto z = from_abi<Fq>(mem_load(reinterpret_cast<const char*>(p) + 64));
    if (fe_is_zero_modp<Fq>(z.v)) return g1j_identity();
    r.z = z;
    return r;
}
ZK_HD __forceinline__ void g1j_store_abi(void* p, const g1j& v) {
    mem_store(p, to_abi(v.x));
    mem_store(reinterpret_cast<char*>(p) + 32, to_abi(v.y));
    mem_store(reinterpret_cast<char*>(p) + 64, to_abi(v.z));
}

// dbl-2009-l (a = 0): 2M + 5S (+2 contractions)
ZK_HD inline g1j g1j_double(const g1j& p) {
    if (g1j_is_id(p)) return p;
    auto a = sqr(p.x);
    auto b = sqr(p.y);
    auto c = sqr(b);
    auto xb = sqr(p.x + b);
    auto d = mul_small<2>(xb - (a + c));
    auto e = mul_small<3>(a);
    auto f = sqr(e);
    g1j r;
    r.z = mul_small<2>(p.y * p.z);
    auto x3 = reduce(f - mul_small<2>(d));
    r.y = reduce(e * (d - x3) - mul_small<8>(c));
    r.x = x3;
    return r;
}

// madd-2007-bl with the exceptional cases (Z3 = 2 Z1 H); q affine with coordinates < 2p: 8M + 3S
ZK_HD inline g1j g1j_add_mixed(const g1j& p, const g1a& q) {
    if (g1a_is_id(q)) return p;
    if (g1j_is_id(p)) return g1j_from_affine(q);
    auto z1z1 = sqr(p.z);
    auto u2 = q.x * z1z1;
    auto s2 = q.y * z1z1 * p.z;
    auto h = u2 - p.x;
    auto rr = s2 - p.y;
    if (is_zero(h)) {
        if (is_zero(rr)) return g1j_double(p);
        return g1j_identity();
    }
    auto hh = sqr(h);
    auto i = mul_small<4>(hh);
    auto j = h * i;
    auto r = mul_small<2>(rr);
    auto v = p.x * i;
    g1j o;
    auto x3 = sqr(r) - (j + mul_small<2>(v));
    o.y = r * (v - x3) - mul_small<2>(p.y * j);
    o.z = mul_small<2>(p.z * h);
    o.x = x3;
    return o;
}

// add-2007-bl with the exceptional cases: 11M + 5S
ZK_HD inline g1j g1j_add(const g1j& p, const g1j& q) {
    if (g1j_is_id(p)) return q;
    if (g1j_is_id(q)) return p;
    auto z1z1 = sqr(p.z);
    auto z2z2 = sqr(q.z);
    auto u1 = p.x * z2z2;
    auto u2 = q.x * z1z1;
    auto s1 = p.y * q.z * z2z2;
    auto s2 = q.y * p.z * z1z1;
    auto h = u2 - u1;
    auto rr = s2 - s1;
    if (is_zero(h)) {
        if (is_zero(rr)) return g1j_double(p);
        return g1j_identity();
    }
    auto i = sqr(mul_small<2>(h));
    auto j = h * i;
    auto r = mul_small<2>(rr);
    auto v = u1 * i;
    g1j o;
    auto x3 = sqr(r) - (j + mul_small<2>(v));
    o.y = r * (v - x3) - mul_small<2>(s1 * j);
    auto zz = sqr(p.z + q.z);
    o.z = (zz - (z1z1 + z2z2)) * h;
    o.x = x3;
    return o;
}

// ---- XYZZ coordinates (x = X / ZZ, y = Y / ZZZ, ZZ^3 = ZZZ^2) for the MSM accumulators: a mixed addition is
// 8M + 2S against 8M + 3S in Jacobian form, a full addition 12M + 2S against 11M + 5S.  ZZ and ZZZ are always
// direct products (< 2p), which also keeps the lazy bounds small.  Identity = exact ZZ = 0.
constexpr int XBX = 8 * U, XBY = 6 * U;
struct g1x { el<Fq, XBX> x; el<Fq, XBY> y; el2<Fq> zz, zzz; };

ZK_HD __forceinline__ bool g1x_is_id(const g1x& p) { return fe_is_zero_exact(p.zz.v); }
ZK_HD __forceinline__ g1x g1x_identity() {
    g1x r;
    r.x = zero<Fq>(); r.y = one<Fq>(); r.zz = zero<Fq>(); r.zzz = zero<Fq>();
    return r;
}
ZK_HD __forceinline__ g1x g1x_from_affine(const g1a& a) {
    if (g1a_is_id(a)) return g1x_identity();
    g1x r;
    r.x = a.x; r.y = a.y; r.zz = one<Fq>(); r.zzz = one<Fq>();
    return r;
}
ZK_HD __forceinline__ g1x g1x_load_raw(const void* p) {
    g1x r;
    const char* c = reinterpret_cast<const char*>(p);
    r.x = load_raw<Fq>(c); r.y = load_raw<Fq>(c + 32); r.zz = load_raw<Fq>(c + 64); r.zzz = load_raw<Fq>(c + 96);
    return r;
}
ZK_HD __forceinline__ void g1x_store_raw(void* p, const g1x& v) {
    char* c = reinterpret_cast<char*>(p);
    store_raw<Fq>(c, v.x); store_raw<Fq>(c + 32, v.y); store_raw<Fq>(c + 64, v.zz); store_raw<Fq>(c + 96, v.zzz);
}
// dbl-2008-s-1 (a = 0): 6M + 3S
ZK_HD inline g1x g1x_double(const g1x& p) {
    if (g1x_is_id(p)) return p;
    auto u = mul_small<2>(p.y);
    auto v = sqr(u);
    auto w = u * v;
    auto s = p.x * v;
    auto m = mul_small<3>(sqr(p.x));
    g1x r;
    auto x3 = sqr(m) - mul_small<2>(s);
    r.y = muladd2(m, s - x3, neg(w), p.y);
    r.zz = v * p.zz;
    r.zzz = w * p.zzz;
    r.x = x3;
    return r;
}
// madd-2008-s with the exceptional cases: 8M + 2S.
// The general formula runs FIRST and unconditionally; the exceptional cases (q = identity, p = identity, p = +-q) are ONE unlikely branch
// behind it that overwrites the result.  Written with early returns in front of the formula, the compiler merged four result paths into
// the accumulator's registers on the hot path of every caller's loop: measured (tools/ab/add_path_bench.hip) 13.1 G additions/s against
// 15.1 for the bare formula; this form: 14.3.  q = identity <=> y = 0 (a prime-order group has no point of order 2; table entries are
// canonical), p = identity <=> zz = 0 (exact zeros), p = +-q needs the three-instruction filter of is_zero before the exact test.
ZK_HD inline g1x g1x_add_mixed(const g1x& p, const g1a& q) {
    auto u2 = q.x * p.zz;
    auto s2 = q.y * p.zzz;
    auto pp_ = u2 - p.x;
    auto r = s2 - p.y;
    auto pp = sqr(pp_);
    auto ppp = pp_ * pp;
    auto q_ = p.x * pp;
    g1x o;
    auto x3 = sqr(r) - add_mul_small<2>(ppp, q_);
    o.y = muladd2(r, sub_lazy(q_, x3), neg_lazy(p.y), ppp);   // r (Q - X3) - Y1 PPP: two products, one reduction, no carry pass on the two differences
    o.zz = p.zz * pp;
    o.zzz = p.zzz * ppp;
    o.x = x3;
    const uint32_t rare = (uint32_t)fe_is_zero_exact(q.y.v) | (uint32_t)g1x_is_id(p) | (uint32_t)maybe_zero(pp_);
    if (__builtin_expect(rare != 0, 0)) {
        if (g1a_is_id(q)) o = p;
        else if (g1x_is_id(p)) o = g1x_from_affine(q);
        else if (is_zero(pp_)) {
            if (is_zero(r)) o = g1x_double(p);
            else o = g1x_identity();
        }
    }
    return o;
}
// add-2008-s with the exceptional cases: 12M + 2S (the same shape: formula first, one unlikely fix-up)
ZK_HD inline g1x g1x_add(const g1x& p, const g1x& q) {
    auto u1 = p.x * q.zz;
    auto u2 = q.x * p.zz;
    auto s1 = p.y * q.zzz;
    auto s2 = q.y * p.zzz;
    auto pp_ = u2 - u1;
    auto r = s2 - s1;
    auto pp = sqr(pp_);
    auto ppp = pp_ * pp;
    auto q_ = u1 * pp;
    g1x o;
    auto x3 = sqr(r) - add_mul_small<2>(ppp, q_);
    o.y = muladd2(r, sub_lazy(q_, x3), neg_lazy(s1), ppp);
    o.zz = p.zz * q.zz * pp;
    o.zzz = p.zzz * q.zzz * ppp;
    o.x = x3;
    const uint32_t rare = (uint32_t)g1x_is_id(p) | (uint32_t)g1x_is_id(q) | (uint32_t)maybe_zero(pp_);
    if (__builtin_expect(rare != 0, 0)) {
        if (g1x_is_id(p)) o = q;
        else if (g1x_is_id(q)) o = p;
        else if (is_zero(pp_)) {
            if (is_zero(r)) o = g1x_double(p);
            else o = g1x_identity();
        }
    }
    return o;
}
// XYZZ -> a Jacobian representative with Z = ZZ * ZZZ (no inversion): X' = X ZZ ZZZ^2, Y' = Y ZZ^3 ZZZ^2
ZK_HD inline g1j g1x_to_jacobian(const g1x& p) {
    if (g1x_is_id(p)) return g1j_identity();
    auto z3sq = sqr(p.zzz);
    auto t = p.zz * z3sq;            // ZZ ZZZ^2
    g1j r;
    r.x = p.x * t;
    r.y = p.y * (t * sqr(p.zz));     // ZZ^3 ZZZ^2
    r.z = p.zz * p.zzz;
    return r;
}

// table entry (canonical coordinates) -> (x, +-y) as a madd operand; the identity stays the exact (0, 0)
ZK_HD __forceinline__ g1a g1a_load_raw_cneg(const void* p, bool do_neg) {
    el1<Fq> x = load_raw<Fq>(p), y = load_raw<Fq>(reinterpret_cast<const char*>(p) + 32);
    bool id = fe_is_zero_exact(x.v) && fe_is_zero_exact(y.v);
    g1a r;
    r.x = x;
    r.y = select(do_neg && !id, neg(y), el2<Fq>(y));   // 2p - y < 2p
    return r;
}
ZK_HD inline g1a g1j_to_affine(const g1j& p) {
    g1a r;
    if (g1j_is_id(p)) return g1a_identity();
    el2<Fq> zi = inv<Fq>(reduce(p.z));
    auto zi2 = sqr(zi);
    r.x = p.x * zi2;
    r.y = p.y * (zi2 * zi);
    return r;
}

// repo-wide synthetic generator (oracle/pyref.py splitmix64 / synth_raw253)
ZK_HD __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
ZK_HD __forceinline__ fe32 synth_raw253(uint64_t seed, uint64_t idx) {
    fe32 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        uint64_t w = splitmix64(seed + (idx * 4 + (uint64_t)j) * 0x2545F4914F6CDD1Dull);
        if (j == 3) w &= 0x1FFFFFFFFFFFFFFFull;
        r.w[2 * j] = (uint32_t)w;
        r.w[2 * j + 1] = (uint32_t)(w >> 32);
    }
    return r;
}

}  // namespace zk
