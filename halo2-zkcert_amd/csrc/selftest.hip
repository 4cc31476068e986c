// selftest.hip — host-side entry points that exercise bn254.hpp exactly as the kernels use it.
// The field / curve code is __host__ __device__, so the CPU suite (tests/test_field_host.py) checks the
// same source the GPU runs, against the oracle, without a GPU.  Not part of include/zkhip.h.
#include "common.hpp"
using namespace zk;

template <class P>
static void t_mul(const uint64_t* a, const uint64_t* b, uint64_t* out) {
    mem_store(out, to_abi(from_abi<P>(mem_load(a)) * from_abi<P>(mem_load(b))));
}
// lazy chain: exercises add / sub / neg / mul_small / x32 loads with growing bounds, then contracts
template <class P>
static void t_chain(const uint64_t* a, const uint64_t* b, uint64_t* out) {
    auto x = load_x32<P>(a);                 // 32 p
    auto y = load_x32<P>(b);                 // 32 p
    auto s = x + y;                          // 64
    auto d = x - y;                          // 65
    auto m = s * d;                          // (x+y)(x-y)
    auto t = mul_small<2>(m) - neg(x) + mul_small<8>(one<P>());   // 2(x^2-y^2) + x + 8   (~110 p)
    auto u = reduce(t * t) - m;              // t^2 - (x^2 - y^2)
    store_div32<P>(out, u);
}
template <class P>
static void t_inv(const uint64_t* a, uint64_t* out) { mem_store(out, to_abi(inv<P>(from_abi<P>(mem_load(a))))); }

extern "C" {
void zkt_fq_mul(const uint64_t* a, const uint64_t* b, uint64_t* o) { t_mul<Fq>(a, b, o); }
void zkt_fr_mul(const uint64_t* a, const uint64_t* b, uint64_t* o) { t_mul<Fr>(a, b, o); }
void zkt_fq_chain(const uint64_t* a, const uint64_t* b, uint64_t* o) { t_chain<Fq>(a, b, o); }
void zkt_fr_chain(const uint64_t* a, const uint64_t* b, uint64_t* o) { t_chain<Fr>(a, b, o); }
void zkt_fq_inv(const uint64_t* a, uint64_t* o) { t_inv<Fq>(a, o); }
void zkt_fr_inv(const uint64_t* a, uint64_t* o) { t_inv<Fr>(a, o); }
void zkt_fq_inv_euclid(const uint64_t* a, uint64_t* o) { mem_store(o, to_abi(inv_euclid<Fq>(from_abi<Fq>(mem_load(a))))); }
void zkt_fr_inv_euclid(const uint64_t* a, uint64_t* o) { mem_store(o, to_abi(inv_euclid<Fr>(from_abi<Fr>(mem_load(a))))); }
void zkt_fq_inv_host(const uint64_t* a, uint64_t* o) { mem_store(o, to_abi(inv_host<Fq>(from_abi<Fq>(mem_load(a))))); }
void zkt_fr_inv_host(const uint64_t* a, uint64_t* o) { mem_store(o, to_abi(inv_host<Fr>(from_abi<Fr>(mem_load(a))))); }
void zkt_fr_raw_roundtrip(const uint64_t* a, uint64_t* o) { store_raw<Fr>(o, load_raw<Fr>(a)); }
void zkt_fr_x32_roundtrip(const uint64_t* a, uint64_t* o) { store_div32<Fr>(o, load_x32<Fr>(a)); }
void zkt_fr_to_canonical(const uint64_t* a, uint64_t* o) { mem_store(o, abi_to_canonical_words<Fr>(mem_load(a))); }
void zkt_fr_from_canonical(const uint64_t* a, uint64_t* o) { mem_store(o, to_abi(from_canonical_words<Fr>((const uint32_t*)a))); }
// the same sum through the XYZZ accumulator the MSM kernels use
void zkt_g1x_sum_mixed(const uint64_t* pts_abi, const uint8_t* negs, size_t n, uint64_t* o) {
    g1x acc = g1x_identity(), other = g1x_identity();
    for (size_t i = 0; i < n; ++i) {
        uint64_t raw[8];
        g1a_store_raw(raw, g1a_load_abi(pts_abi + 8 * i));
        g1x& tgt = (i & 1) ? other : acc;          // two partial sums, folded with the full addition below
        tgt = g1x_add_mixed(tgt, g1a_load_raw_cneg(raw, negs[i] != 0));
        if (i % 5 == 2) {
            uint64_t x[16];
            g1x_store_raw(x, tgt);
            tgt = g1x_load_raw(x);
        }
        if (i % 64 == 63) { acc = g1x_add(acc, other); other = g1x_identity(); }
    }
    acc = g1x_add(acc, other);
    acc = g1x_add(acc, g1x_double(acc));            // 3 * sum
    g1j_store_abi(o, g1x_to_jacobian(acc));
}
// canonical(v + k p) == v for every k the lazy representation allows
int zkt_canon_kp(const uint64_t* a, uint32_t k, int field) {
    fe v = fe_split<0>(mem_load(a)), s = v;
    uint64_t c = 0;
    for (int i = 0; i < 9; ++i) {
        c += (uint64_t)v.l[i] + (uint64_t)k * (field ? FrP::M[i] : FqP::M[i]);
        s.l[i] = i < 8 ? (uint32_t)c & LMASK : (uint32_t)c;
        c >>= LB;
    }
    fe r = field ? fe_canonical<Fr>(s) : fe_canonical<Fq>(s);
    int ok = 1;
    for (int i = 0; i < 9; ++i) ok &= r.l[i] == v.l[i];
    return ok;
}
void zkt_g1_add_mixed(const uint64_t* p, const uint64_t* q, uint64_t* o) { g1j_store_abi(o, g1j_add_mixed(g1j_load_abi(p), g1a_load_abi(q))); }
void zkt_g1_double(const uint64_t* p, uint64_t* o) { g1j_store_abi(o, g1j_double(g1j_load_abi(p))); }
// acc = sum_{i < n} (neg_i ? -pts[i] : pts[i]) with mixed adds, through the raw (R') storage format
void zkt_g1_sum_mixed(const uint64_t* pts_abi, const uint8_t* negs, size_t n, uint64_t* o) {
    g1j acc = g1j_identity();
    for (size_t i = 0; i < n; ++i) {
        uint64_t raw[8];
        g1a_store_raw(raw, g1a_load_abi(pts_abi + 8 * i));
        acc = g1j_add_mixed(acc, g1a_load_raw_cneg(raw, negs[i] != 0));
        if (i % 7 == 3) {  // round-trip the accumulator through the scratch format too
            uint64_t j[12];
            g1j_store_raw(j, acc);
            acc = g1j_load_raw(j);
        }
    }
    g1j_store_abi(o, acc);
}
}
