// bn254.hpp — device-side BN254 Fq / Fr Montgomery arithmetic and G1 group law for gfx950.
//
// Replaces (on the device) halo2curves 0.4.0 src/bn256/{fq,fr,curve}.rs + src/derive/{field,curve}.rs
// [UPSTREAM-RECALL; crate pinned at /root/reference/Cargo.lock:1359-1361].  Same memory layout as
// the Rust types: 4 little-endian u64 limbs in Montgomery form (R = 2^256), here viewed as 8 u32
// limbs because the CDNA4 integer multiplier is 32x32 (v_mad_u64_u32).
//
// No MFMA: none of this is a dense contraction.  The cost model is the 32-bit multiply pipe:
// one Montgomery product = 8*8 + 8*8 + 8 = 136 v_mad_u64_u32 per lane.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ZK_HD __host__ __device__

namespace zk {

struct alignas(16) fe {
    uint32_t l[8];
};

struct FqP {
    static constexpr uint32_t M[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                                      0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr uint32_t INV = 0xe4866389u;  // -M^-1 mod 2^32
    static constexpr uint32_t ONE[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u,
                                        0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    static constexpr uint32_t R2[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u,
                                       0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
};
struct FrP {
    static constexpr uint32_t M[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u,
                                      0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr uint32_t INV = 0xefffffffu;
    static constexpr uint32_t ONE[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u,
                                        0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    static constexpr uint32_t R2[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u,
                                       0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
};

ZK_HD __forceinline__ fe fe_zero() {
    fe r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.l[i] = 0;
    return r;
}
template <class P>
ZK_HD __forceinline__ fe fe_one() {
    fe r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.l[i] = P::ONE[i];
    return r;
}
ZK_HD __forceinline__ bool fe_is_zero(const fe& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) o |= a.l[i];
    return o == 0;
}
ZK_HD __forceinline__ bool fe_eq(const fe& a, const fe& b) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) o |= a.l[i] ^ b.l[i];
    return o == 0;
}

// 16-byte vector loads/stores (coalesced: lane i touches element i).
ZK_HD __forceinline__ fe fe_load(const void* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 a = q[0], b = q[1];
    fe r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    return r;
}
ZK_HD __forceinline__ void fe_store(void* p, const fe& v) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

// r = (t >= M) ? t - M : t, branch-free.
template <class P>
ZK_HD __forceinline__ void fe_cond_sub(fe& t) {
    uint32_t d[8];
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t x = (uint64_t)t.l[i] - P::M[i] - br;
        d[i] = (uint32_t)x;
        br = (x >> 32) & 1;
    }
    bool keep = br != 0;  // t < M
#pragma unroll
    for (int i = 0; i < 8; ++i) t.l[i] = keep ? t.l[i] : d[i];
}

template <class P>
ZK_HD __forceinline__ fe fe_add(const fe& a, const fe& b) {
    fe t;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        c += (uint64_t)a.l[i] + b.l[i];
        t.l[i] = (uint32_t)c;
        c >>= 32;
    }
    fe_cond_sub<P>(t);  // both moduli < 2^254: a + b < 2^255, no carry out
    return t;
}
template <class P>
ZK_HD __forceinline__ fe fe_sub(const fe& a, const fe& b) {
    fe t;
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t x = (uint64_t)a.l[i] - b.l[i] - br;
        t.l[i] = (uint32_t)x;
        br = (x >> 32) & 1;
    }
    uint32_t mask = br ? 0xffffffffu : 0u;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        c += (uint64_t)t.l[i] + (P::M[i] & mask);
        t.l[i] = (uint32_t)c;
        c >>= 32;
    }
    return t;
}
template <class P>
ZK_HD __forceinline__ fe fe_neg(const fe& a) {
    return fe_sub<P>(fe_zero(), a);
}
template <class P>
ZK_HD __forceinline__ fe fe_dbl(const fe& a) {
    return fe_add<P>(a, a);
}

// CIOS Montgomery product, 8 x 32-bit limbs.  M < 2^254 so the running value stays below
// 2^33 * M < 2^288: nine limbs suffice.
template <class P>
ZK_HD __forceinline__ fe fe_mul(const fe& a, const fe& b) {
    uint32_t t[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            c = (uint64_t)a.l[j] * b.l[i] + t[j] + c;
            t[j] = (uint32_t)c;
            c >>= 32;
        }
        t[8] += (uint32_t)c;
        uint32_t q = t[0] * P::INV;
        c = (uint64_t)q * P::M[0] + t[0];
        c >>= 32;
#pragma unroll
        for (int j = 1; j < 8; ++j) {
            c = (uint64_t)q * P::M[j] + t[j] + c;
            t[j - 1] = (uint32_t)c;
            c >>= 32;
        }
        c += t[8];
        t[7] = (uint32_t)c;
        t[8] = (uint32_t)(c >> 32);
    }
    fe r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.l[i] = t[i];
    fe_cond_sub<P>(r);
    return r;
}
template <class P>
ZK_HD __forceinline__ fe fe_sqr(const fe& a) {
    return fe_mul<P>(a, a);
}
template <class P>
ZK_HD __forceinline__ fe fe_from_mont(const fe& a) {
    fe one = fe_zero();
    one.l[0] = 1;
    return fe_mul<P>(a, one);
}
template <class P>
ZK_HD __forceinline__ fe fe_to_mont(const fe& a) {
    fe r2;
#pragma unroll
    for (int i = 0; i < 8; ++i) r2.l[i] = P::R2[i];
    return fe_mul<P>(a, r2);
}
// a^e, e given as 8 u32 limbs (runtime), left-to-right.
template <class P>
ZK_HD inline fe fe_pow(const fe& a, const uint32_t e[8]) {
    fe acc = fe_one<P>();
    bool started = false;
    for (int i = 255; i >= 0; --i) {
        if (started) acc = fe_sqr<P>(acc);
        if ((e[i >> 5] >> (i & 31)) & 1) {
            acc = started ? fe_mul<P>(acc, a) : a;
            started = true;
        }
    }
    return acc;
}
template <class P>
ZK_HD inline fe fe_pow_u64(const fe& a, uint64_t e) {
    fe acc = fe_one<P>();
    bool started = false;
    for (int i = 63; i >= 0; --i) {
        if (started) acc = fe_sqr<P>(acc);
        if ((e >> i) & 1) {
            acc = started ? fe_mul<P>(acc, a) : a;
            started = true;
        }
    }
    return acc;
}
// Fermat inverse (0 -> 0).
template <class P>
ZK_HD inline fe fe_inv(const fe& a) {
    uint32_t e[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = P::M[i];
    e[0] -= 2;
    if (fe_is_zero(a)) return a;
    return fe_pow<P>(a, e);
}

using Fq = FqP;
using Fr = FrP;

// Fr constants (canonical limbs; halo2curves src/bn256/fr.rs): 2^28-th root of unity, ZETA, DELTA.
constexpr uint32_t FR_S = 28;
constexpr uint32_t FR_ROOT_OF_UNITY[8] = {0x60c37c9cu, 0xd34f1ed9u, 0xd39329c8u, 0x3215cf6du,
                                          0x3dd31f74u, 0x98865ea9u, 0x166d18b7u, 0x03ddb9f5u};
constexpr uint32_t FR_ZETA[8] = {0xb99c90ddu, 0x8b17ea66u, 0x8d8daaa7u, 0x5bfc4108u, 0x41a91758u, 0xb3c4d79du, 0u, 0u};
constexpr uint32_t FR_DELTA[8] = {0xe533e9a2u, 0x870e56bbu, 0x5e963f25u, 0x5b5f898eu,
                                  0xd4c86e71u, 0x64ec26aau, 0x22c6f0cau, 0x09226b6eu};
template <class P>
ZK_HD __forceinline__ fe fe_from_canonical(const uint32_t v[8]) {
    fe t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t.l[i] = v[i];
    return fe_to_mont<P>(t);
}
template <class P>
ZK_HD __forceinline__ fe fe_from_u64(uint64_t v) {
    fe t = fe_zero();
    t.l[0] = (uint32_t)v;
    t.l[1] = (uint32_t)(v >> 32);
    return fe_to_mont<P>(t);
}

// ------------------------------------------------------------------------------------ G1
// y^2 = x^3 + 3.  Affine identity = (0,0) (halo2curves G1Affine); Jacobian identity z = 0.
struct g1a { fe x, y; };
struct g1j { fe x, y, z; };

ZK_HD __forceinline__ bool g1a_is_id(const g1a& p) { return fe_is_zero(p.x) && fe_is_zero(p.y); }
ZK_HD __forceinline__ bool g1j_is_id(const g1j& p) { return fe_is_zero(p.z); }
ZK_HD __forceinline__ g1j g1j_identity() {
    g1j r;
    r.x = fe_zero(); r.y = fe_one<Fq>(); r.z = fe_zero();
    return r;
}
ZK_HD __forceinline__ g1j g1j_from_affine(const g1a& a) {
    if (g1a_is_id(a)) return g1j_identity();
    g1j r;
    r.x = a.x; r.y = a.y; r.z = fe_one<Fq>();
    return r;
}
ZK_HD __forceinline__ g1a g1a_load(const void* p) {
    g1a r;
    r.x = fe_load(p);
    r.y = fe_load(reinterpret_cast<const char*>(p) + 32);
    return r;
}
ZK_HD __forceinline__ void g1a_store(void* p, const g1a& v) {
    fe_store(p, v.x);
    fe_store(reinterpret_cast<char*>(p) + 32, v.y);
}
ZK_HD __forceinline__ g1j g1j_load(const void* p) {
    g1j r;
    r.x = fe_load(p);
    r.y = fe_load(reinterpret_cast<const char*>(p) + 32);
    r.z = fe_load(reinterpret_cast<const char*>(p) + 64);
    return r;
}
ZK_HD __forceinline__ void g1j_store(void* p, const g1j& v) {
    fe_store(p, v.x);
    fe_store(reinterpret_cast<char*>(p) + 32, v.y);
    fe_store(reinterpret_cast<char*>(p) + 64, v.z);
}

// dbl-2009-l (a = 0): 2M + 5S
ZK_HD inline g1j g1j_double(const g1j& p) {
    if (g1j_is_id(p)) return p;
    fe a = fe_sqr<Fq>(p.x);
    fe b = fe_sqr<Fq>(p.y);
    fe c = fe_sqr<Fq>(b);
    fe d = fe_add<Fq>(p.x, b);
    d = fe_sqr<Fq>(d);
    d = fe_sub<Fq>(fe_sub<Fq>(d, a), c);
    d = fe_dbl<Fq>(d);
    fe e = fe_add<Fq>(fe_dbl<Fq>(a), a);
    fe f = fe_sqr<Fq>(e);
    g1j r;
    r.z = fe_dbl<Fq>(fe_mul<Fq>(p.y, p.z));
    r.x = fe_sub<Fq>(f, fe_dbl<Fq>(d));
    fe c8 = fe_dbl<Fq>(fe_dbl<Fq>(fe_dbl<Fq>(c)));
    r.y = fe_sub<Fq>(fe_mul<Fq>(e, fe_sub<Fq>(d, r.x)), c8);
    return r;
}

// madd-2007-bl with exceptional cases: 7M + 4S
ZK_HD inline g1j g1j_add_mixed(const g1j& p, const g1a& q) {
    if (g1a_is_id(q)) return p;
    if (g1j_is_id(p)) return g1j_from_affine(q);
    fe z1z1 = fe_sqr<Fq>(p.z);
    fe u2 = fe_mul<Fq>(q.x, z1z1);
    fe s2 = fe_mul<Fq>(fe_mul<Fq>(q.y, z1z1), p.z);
    if (fe_eq(p.x, u2)) {
        if (fe_eq(p.y, s2)) return g1j_double(p);
        return g1j_identity();
    }
    fe h = fe_sub<Fq>(u2, p.x);
    fe hh = fe_sqr<Fq>(h);
    fe i = fe_dbl<Fq>(fe_dbl<Fq>(hh));
    fe j = fe_mul<Fq>(h, i);
    fe r = fe_dbl<Fq>(fe_sub<Fq>(s2, p.y));
    fe v = fe_mul<Fq>(p.x, i);
    g1j o;
    o.x = fe_sub<Fq>(fe_sub<Fq>(fe_sub<Fq>(fe_sqr<Fq>(r), j), v), v);
    fe yj = fe_dbl<Fq>(fe_mul<Fq>(p.y, j));
    o.y = fe_sub<Fq>(fe_mul<Fq>(r, fe_sub<Fq>(v, o.x)), yj);
    fe zh = fe_add<Fq>(p.z, h);
    o.z = fe_sub<Fq>(fe_sub<Fq>(fe_sqr<Fq>(zh), z1z1), hh);
    return o;
}

// add-2007-bl with exceptional cases: 11M + 5S
ZK_HD inline g1j g1j_add(const g1j& p, const g1j& q) {
    if (g1j_is_id(p)) return q;
    if (g1j_is_id(q)) return p;
    fe z1z1 = fe_sqr<Fq>(p.z);
    fe z2z2 = fe_sqr<Fq>(q.z);
    fe u1 = fe_mul<Fq>(p.x, z2z2);
    fe u2 = fe_mul<Fq>(q.x, z1z1);
    fe s1 = fe_mul<Fq>(fe_mul<Fq>(p.y, q.z), z2z2);
    fe s2 = fe_mul<Fq>(fe_mul<Fq>(q.y, p.z), z1z1);
    if (fe_eq(u1, u2)) {
        if (fe_eq(s1, s2)) return g1j_double(p);
        return g1j_identity();
    }
    fe h = fe_sub<Fq>(u2, u1);
    fe i = fe_sqr<Fq>(fe_dbl<Fq>(h));
    fe j = fe_mul<Fq>(h, i);
    fe r = fe_dbl<Fq>(fe_sub<Fq>(s2, s1));
    fe v = fe_mul<Fq>(u1, i);
    g1j o;
    o.x = fe_sub<Fq>(fe_sub<Fq>(fe_sub<Fq>(fe_sqr<Fq>(r), j), v), v);
    fe sj = fe_dbl<Fq>(fe_mul<Fq>(s1, j));
    o.y = fe_sub<Fq>(fe_mul<Fq>(r, fe_sub<Fq>(v, o.x)), sj);
    fe zz = fe_add<Fq>(p.z, q.z);
    o.z = fe_mul<Fq>(fe_sub<Fq>(fe_sub<Fq>(fe_sqr<Fq>(zz), z1z1), z2z2), h);
    return o;
}

ZK_HD __forceinline__ g1a g1a_neg(const g1a& p) {
    g1a r;
    r.x = p.x;
    r.y = fe_is_zero(p.y) ? p.y : fe_sub<Fq>(fe_zero(), p.y);
    return r;
}
ZK_HD __forceinline__ g1a g1a_cneg(const g1a& p, bool neg) {
    g1a n = g1a_neg(p);
    g1a r;
#pragma unroll
    for (int i = 0; i < 8; ++i) { r.x.l[i] = p.x.l[i]; r.y.l[i] = neg ? n.y.l[i] : p.y.l[i]; }
    return r;
}
ZK_HD inline g1a g1j_to_affine(const g1j& p) {
    g1a r;
    if (g1j_is_id(p)) { r.x = fe_zero(); r.y = fe_zero(); return r; }
    fe zi = fe_inv<Fq>(p.z);
    fe zi2 = fe_sqr<Fq>(zi);
    r.x = fe_mul<Fq>(p.x, zi2);
    r.y = fe_mul<Fq>(p.y, fe_mul<Fq>(zi2, zi));
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
ZK_HD __forceinline__ fe synth_raw253(uint64_t seed, uint64_t idx) {
    fe r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        uint64_t w = splitmix64(seed + (idx * 4 + (uint64_t)j) * 0x2545F4914F6CDD1Dull);
        if (j == 3) w &= 0x1FFFFFFFFFFFFFFFull;
        r.l[2 * j] = (uint32_t)w;
        r.l[2 * j + 1] = (uint32_t)(w >> 32);
    }
    return r;
}

}  // namespace zk
