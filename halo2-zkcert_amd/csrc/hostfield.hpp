// hostfield.hpp — Fr values on the host (R' form, canonical) for the few-hundred field operations the prover's host side
// performs between launches (interpolation, vanishing products, challenge powers).  Included by shplonk.hip and prover.hip.
#pragma once
#include <vector>

#include "bn254.hpp"

namespace zk {
struct HF { fe v; };
inline HF hf(const el1<Fr>& e) { return HF{e.v}; }
inline el1<Fr> E(const HF& a) { return el1<Fr>(a.v); }
inline HF hmul(const HF& a, const HF& b) { return hf(canonical(E(a) * E(b))); }
inline HF hadd(const HF& a, const HF& b) { return hf(canonical(E(a) + E(b))); }
inline HF hsub(const HF& a, const HF& b) { return hf(canonical(E(a) - E(b))); }
inline HF hzero() { return hf(zero<Fr>()); }
inline HF hone() { return hf(one<Fr>()); }
inline HF hf_from_abi(const uint64_t* p) { return hf(canonical(from_abi<Fr>(mem_load(p)))); }
inline fe32 hf_raw(const HF& a) { return fe_pack(a.v); }                 // what the kernels take as a scalar
inline fe32 hf_abi(const HF& a) { return to_abi(E(a)); }                 // polynomial-coefficient scale
inline bool hf_is_zero(const HF& a) { return fe_is_zero_exact(a.v); }
// one inversion for the whole list (Montgomery's trick); zeros are not expected
inline void hf_batch_invert(std::vector<HF>& xs) {
    std::vector<HF> pre(xs.size());
    HF acc = hone();
    for (size_t i = 0; i < xs.size(); ++i) { pre[i] = acc; acc = hmul(acc, xs[i]); }
    HF iv = hf(canonical(inv_host<Fr>(el2<Fr>(E(acc)))));
    for (size_t i = xs.size(); i-- > 0;) { HF t = hmul(iv, pre[i]); iv = hmul(iv, xs[i]); xs[i] = t; }
}
struct Words { uint32_t w[8]; };
inline Words canon_words(const HF& a) { fe32 m = to_canonical_words(E(a)); Words r; for (int i = 0; i < 8; ++i) r.w[i] = m.w[i]; return r; }
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
