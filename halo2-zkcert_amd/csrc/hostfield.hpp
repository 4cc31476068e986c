// hostfield.hpp — Fr values on the host for the few hundred field operations the prover's host side performs between
// launches (interpolation, vanishing products, challenge powers): they sit on the critical path after every Fiat-Shamir
// round trip, so they use the host's native arithmetic — 4 x 64-bit Montgomery (R = 2^256, canonical: bit-identical to
// the ABI form) with 128-bit products, ~10x the speed of the device-oriented 29-bit limb code compiled for the host.
// Included by shplonk.hip and prover.hip.
#pragma once
#include <vector>

#include "bn254.hpp"

namespace zk {
struct HF { uint64_t w[4]; };

// generic 4 x 64-bit Montgomery core (modulus P < 2^254, INV = -P^-1 mod 2^64); values canonical
namespace hostmont {
typedef unsigned __int128 u128;
inline bool geq(const uint64_t* a, const uint64_t* P) {
    for (int i = 3; i >= 0; --i) if (a[i] != P[i]) return a[i] > P[i];
    return true;
}
inline void sub_mod(uint64_t* a, const uint64_t* P) {
    u128 br = 0;
    for (int i = 0; i < 4; ++i) { u128 d = (u128)a[i] - P[i] - (uint64_t)br; a[i] = (uint64_t)d; br = (d >> 64) & 1; }
}
inline void mul(uint64_t* r, const uint64_t* a, const uint64_t* b, const uint64_t* P, uint64_t INV) {   // CIOS: a * b / 2^256 mod P
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        u128 c = 0;
        for (int j = 0; j < 4; ++j) { c += (u128)a[j] * b[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * INV;
        c = ((u128)m * P[0] + t[0]) >> 64;
        for (int j = 1; j < 4; ++j) { c += (u128)m * P[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    for (int i = 0; i < 4; ++i) r[i] = t[i];
    if (t[4] || geq(r, P)) sub_mod(r, P);
}
// a * b as 8 limbs (no reduction), t += that, and the Montgomery reduction of 8 limbs: a dot product of k terms is k wide products and
// ONE reduction (valid while the sum stays below 2^512 and the result below 2 P: up to 3 products of canonical BN254 elements)
inline void mul_wide_acc(uint64_t t[8], const uint64_t* a, const uint64_t* b) {
    u128 carry_out = 0;
    for (int i = 0; i < 4; ++i) {
        u128 c = 0;
        for (int j = 0; j < 4; ++j) { c += (u128)a[j] * b[i] + t[i + j]; t[i + j] = (uint64_t)c; c >>= 64; }
        for (int k = i + 4; k < 8 && c; ++k) { c += t[k]; t[k] = (uint64_t)c; c >>= 64; }
        carry_out += c;
    }
    (void)carry_out;
}
inline void reduce_wide(uint64_t* r, uint64_t t[8], const uint64_t* P, uint64_t INV) {
    uint64_t top = 0;
    for (int i = 0; i < 4; ++i) {
        const uint64_t m = t[i] * INV;
        u128 c = 0;
        for (int j = 0; j < 4; ++j) { c += (u128)m * P[j] + t[i + j]; t[i + j] = (uint64_t)c; c >>= 64; }
        for (int k = i + 4; k < 8; ++k) { c += t[k]; t[k] = (uint64_t)c; c >>= 64; }
        top += (uint64_t)c;
    }
    for (int i = 0; i < 4; ++i) r[i] = t[4 + i];
    if (top || geq(r, P)) sub_mod(r, P);
    if (geq(r, P)) sub_mod(r, P);
}
inline void add(uint64_t* r, const uint64_t* a, const uint64_t* b, const uint64_t* P) {
    u128 c = 0;
    for (int i = 0; i < 4; ++i) { c += (u128)a[i] + b[i]; r[i] = (uint64_t)c; c >>= 64; }
    if (c || geq(r, P)) sub_mod(r, P);   // P < 2^254: no carry out in fact
}
inline void sub(uint64_t* r, const uint64_t* a, const uint64_t* b, const uint64_t* P) {
    u128 br = 0;
    for (int i = 0; i < 4; ++i) { u128 d = (u128)a[i] - b[i] - (uint64_t)br; r[i] = (uint64_t)d; br = (d >> 64) & 1; }
    if (br) { u128 c = 0; for (int i = 0; i < 4; ++i) { c += (u128)r[i] + P[i]; r[i] = (uint64_t)c; c >>= 64; } }
}
}  // namespace hostmont

namespace hostfr {
constexpr uint64_t P[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
constexpr uint64_t INV = 0xc2e1f593efffffffull;   // -p^-1 mod 2^64
constexpr uint64_t ONE[4] = {0xac96341c4ffffffbull, 0x36fc76959f60cd29ull, 0x666ea36f7879462eull, 0x0e0a77c19a07df2full};   // 2^256 mod p
inline bool geq_p(const uint64_t* a) { return hostmont::geq(a, P); }
inline void sub_p(uint64_t* a) { hostmont::sub_mod(a, P); }
}  // namespace hostfr
namespace hostfq {
constexpr uint64_t P[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
constexpr uint64_t INV = 0x87d20782e4866389ull;
constexpr uint64_t ONE[4] = {0xd35d438dc58f0d9dull, 0x0a78eb28f5c70b3dull, 0x666ea36f7879462cull, 0x0e0a77c19a07df2full};
}  // namespace hostfq

inline HF hmul(const HF& a, const HF& b) { HF r; hostmont::mul(r.w, a.w, b.w, hostfr::P, hostfr::INV); return r; }
inline HF hadd(const HF& a, const HF& b) { HF r; hostmont::add(r.w, a.w, b.w, hostfr::P); return r; }
inline HF hsub(const HF& a, const HF& b) { HF r; hostmont::sub(r.w, a.w, b.w, hostfr::P); return r; }
// a0 b0 + a1 b1 + a2 b2 with one reduction
inline HF hdot3(const HF& a0, const HF& b0, const HF& a1, const HF& b1, const HF& a2, const HF& b2) {
    uint64_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    hostmont::mul_wide_acc(t, a0.w, b0.w);
    hostmont::mul_wide_acc(t, a1.w, b1.w);
    hostmont::mul_wide_acc(t, a2.w, b2.w);
    HF r;
    hostmont::reduce_wide(r.w, t, hostfr::P, hostfr::INV);
    return r;
}
inline HF hzero() { return HF{{0, 0, 0, 0}}; }
inline HF hone() { return HF{{hostfr::ONE[0], hostfr::ONE[1], hostfr::ONE[2], hostfr::ONE[3]}}; }
inline HF hf_from_abi(const uint64_t* p) {   // the ABI form is this form; callers' values are canonical, reduce anyway
    HF r{{p[0], p[1], p[2], p[3]}};
    while (hostfr::geq_p(r.w)) hostfr::sub_p(r.w);
    return r;
}
inline fe32 hf_words(const HF& a) {
    fe32 m;
    for (int i = 0; i < 4; ++i) { m.w[2 * i] = (uint32_t)a.w[i]; m.w[2 * i + 1] = (uint32_t)(a.w[i] >> 32); }
    return m;
}
inline fe32 hf_abi(const HF& a) { return hf_words(a); }                  // polynomial-coefficient scale
inline fe32 hf_raw(const HF& a) {                                        // what the kernels take as a scalar: x 2^261, canonical
    HF r = a;
    for (int i = 0; i < 5; ++i) r = hadd(r, r);
    return hf_words(r);
}
inline HF hf_from_fe32(const fe32& o) {
    HF r;
    for (int i = 0; i < 4; ++i) r.w[i] = o.w[2 * i] | ((uint64_t)o.w[2 * i + 1] << 32);
    return r;
}
// any 256-bit integer (little-endian words), reduced mod r, into Montgomery form (through the limb layer: transcripts only)
inline HF hf_from_canonical_words(const uint32_t w[8]) { return hf_from_fe32(to_abi(from_canonical_words<Fr>(w))); }
inline bool hf_is_zero(const HF& a) { return (a.w[0] | a.w[1] | a.w[2] | a.w[3]) == 0; }
inline HF hf_invert(const HF& a) {   // binary Euclid of the limb layer (~2.5 us); a != 0
    fe32 m = hf_words(a);
    return hf_from_fe32(to_abi(inv_host<Fr>(from_abi<Fr>(m))));
}
// one inversion for the whole list (Montgomery's trick); zeros are not expected
inline void hf_batch_invert(std::vector<HF>& xs) {
    std::vector<HF> pre(xs.size());
    HF acc = hone();
    for (size_t i = 0; i < xs.size(); ++i) { pre[i] = acc; acc = hmul(acc, xs[i]); }
    HF iv = hf_invert(acc);
    for (size_t i = xs.size(); i-- > 0;) { HF t = hmul(iv, pre[i]); iv = hmul(iv, xs[i]); xs[i] = t; }
}
struct Words { uint32_t w[8]; };
inline Words canon_words(const HF& a) {   // the integer itself: out of Montgomery form
    fe32 m = hf_words(hmul(a, HF{{1, 0, 0, 0}}));
    Words r;
    for (int i = 0; i < 8; ++i) r.w[i] = m.w[i];
    return r;
}
inline bool words_less(const Words& a, const Words& b) {
    for (int i = 7; i >= 0; --i) if (a.w[i] != b.w[i]) return a.w[i] < b.w[i];
    return false;
}
inline HF hpow(HF a, uint64_t e) {
    HF r = hone();
    while (e) { if (e & 1) r = hmul(r, a); a = hmul(a, a); e >>= 1; }
    return r;
}
}  // namespace zk
