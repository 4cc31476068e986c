// transcript.hip — halo2_proofs transcript.rs Blake2bWrite<Vec<u8>, G1Affine, Challenge255<G1Affine>> on the host
// [UPSTREAM-RECALL; crate pinned at /root/reference/Cargo.lock:1320-1322], as a ready-made zk_transcript for zkhip_create_proof:
// BLAKE2b-512 personalised "Halo2-Transcript"; write_point absorbs prefix 1 and the canonical x, y (32 little-endian bytes each)
// and appends the 32-byte compressed point to the proof; write_scalar absorbs prefix 2 and the canonical scalar and appends
// its 32 bytes; a challenge absorbs prefix 0 and is the 64-byte digest of a copy of the state, reduced mod r (little-endian).
// (The reference's own commands use snark-verifier's Poseidon / Keccak transcripts; a Rust caller passes its own callbacks.)
#include <vector>

#include "common.hpp"
#include "hostfield.hpp"
using namespace zk;

namespace {
// BLAKE2b (RFC 7693), unkeyed, 64-byte digest, with a personalisation string
struct Blake2b {
    uint64_t h[8], t0 = 0, t1 = 0;
    uint8_t buf[128];
    size_t buflen = 0;
    static constexpr uint64_t IV[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                                       0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
    void init(const char personal[16]) {
        uint8_t P[64] = {0};
        P[0] = 64; P[2] = 1; P[3] = 1;   // digest length, key length 0, fanout 1, depth 1
        memcpy(P + 48, personal, 16);
        for (int i = 0; i < 8; ++i) { uint64_t w; memcpy(&w, P + 8 * i, 8); h[i] = IV[i] ^ w; }
        t0 = t1 = 0; buflen = 0;
    }
    static inline uint64_t rotr(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
    void compress(const uint8_t block[128], bool last) {
        static const uint8_t S[12][16] = {
            {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
            {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
            {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
            {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
            {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
            {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
        uint64_t m[16], v[16];
        for (int i = 0; i < 16; ++i) memcpy(&m[i], block + 8 * i, 8);
        for (int i = 0; i < 8; ++i) { v[i] = h[i]; v[i + 8] = IV[i]; }
        v[12] ^= t0; v[13] ^= t1;
        if (last) v[14] = ~v[14];
        auto G = [&](int a, int b, int c, int d, uint64_t x, uint64_t y) {
            v[a] = v[a] + v[b] + x; v[d] = rotr(v[d] ^ v[a], 32);
            v[c] = v[c] + v[d];     v[b] = rotr(v[b] ^ v[c], 24);
            v[a] = v[a] + v[b] + y; v[d] = rotr(v[d] ^ v[a], 16);
            v[c] = v[c] + v[d];     v[b] = rotr(v[b] ^ v[c], 63);
        };
        for (int r = 0; r < 12; ++r) {
            const uint8_t* s = S[r];
            G(0, 4, 8, 12, m[s[0]], m[s[1]]);   G(1, 5, 9, 13, m[s[2]], m[s[3]]);
            G(2, 6, 10, 14, m[s[4]], m[s[5]]);  G(3, 7, 11, 15, m[s[6]], m[s[7]]);
            G(0, 5, 10, 15, m[s[8]], m[s[9]]);  G(1, 6, 11, 12, m[s[10]], m[s[11]]);
            G(2, 7, 8, 13, m[s[12]], m[s[13]]); G(3, 4, 9, 14, m[s[14]], m[s[15]]);
        }
        for (int i = 0; i < 8; ++i) h[i] ^= v[i] ^ v[i + 8];
    }
    void update(const uint8_t* in, size_t len) {
        while (len) {
            if (buflen == 128) {   // the buffer is only compressed when more input follows (the last block is special)
                t0 += 128; if (t0 < 128) ++t1;
                compress(buf, false);
                buflen = 0;
            }
            size_t take = std::min(len, (size_t)128 - buflen);
            memcpy(buf + buflen, in, take);
            buflen += take; in += take; len -= take;
        }
    }
    void digest(uint8_t out[64]) const {   // of a copy: the state stays usable
        Blake2b c = *this;
        c.t0 += c.buflen; if (c.t0 < c.buflen) ++c.t1;
        memset(c.buf + c.buflen, 0, 128 - c.buflen);
        c.compress(c.buf, true);
        memcpy(out, c.h, 64);
    }
};
constexpr uint64_t Blake2b::IV[8];
}  // namespace

struct zkhip_blake2b_transcript {
    Blake2b state;
    zk_transcript cb;
    std::vector<uint8_t> proof;          // what Blake2bWrite's writer received
    std::vector<uint64_t> points_xy;     // every written point as affine ABI limbs (8 per point), for callers that want coordinates
    std::vector<uint64_t> challenges;    // every squeezed challenge, ABI limbs (4 per challenge)
};

namespace {
void t_write_point(void* user, const uint8_t bytes32[32], const uint64_t xy[8]) {
    auto* t = (zkhip_blake2b_transcript*)user;
    uint8_t msg[65];
    msg[0] = 1;
    fe32 x = abi_to_canonical_words<Fq>(mem_load(xy)), y = abi_to_canonical_words<Fq>(mem_load(xy + 4));
    memcpy(msg + 1, x.w, 32);
    memcpy(msg + 33, y.w, 32);
    t->state.update(msg, 65);
    t->proof.insert(t->proof.end(), bytes32, bytes32 + 32);
    t->points_xy.insert(t->points_xy.end(), xy, xy + 8);
}
void t_write_scalar(void* user, const uint64_t scalar[4]) {
    auto* t = (zkhip_blake2b_transcript*)user;
    uint8_t msg[33];
    msg[0] = 2;
    fe32 s = abi_to_canonical_words<Fr>(mem_load(scalar));
    memcpy(msg + 1, s.w, 32);
    t->state.update(msg, 33);
    t->proof.insert(t->proof.end(), msg + 1, msg + 33);
}
void t_squeeze(void* user, uint64_t out[4]) {
    auto* t = (zkhip_blake2b_transcript*)user;
    const uint8_t zero = 0;
    t->state.update(&zero, 1);
    uint8_t d[64];
    t->state.digest(d);
    // the 512-bit little-endian integer mod r: hi * 2^256 + lo, both halves as canonical-word inputs (from_canonical_words reduces)
    uint32_t lo[8], hi[8];
    memcpy(lo, d, 32);
    memcpy(hi, d + 32, 32);
    const HF two256 = hf_from_canonical_words(hi);               // hi (< 2^256, reduced by the product with R'^2)
    const HF low = hf_from_canonical_words(lo);
    static const HF r256 = [] {                                                  // 2^256 mod r
        HF v = hone();
        for (int i = 0; i < 256; ++i) v = hadd(v, v);
        return v;
    }();
    const HF c = hadd(hmul(two256, r256), low);
    fe32 abi = hf_abi(c);
    memcpy(out, abi.w, 32);
    t->challenges.insert(t->challenges.end(), out, out + 4);
}
}  // namespace

// ------------------------------------------------------------------ EVM transcript
// snark-verifier system/halo2/transcript/evm.rs EvmTranscript<G1Affine, NativeLoader, _, Vec<u8>> [UPSTREAM-RECALL; crate pinned at
// /root/reference/Cargo.lock:2714-2716; the transcript behind gen_evm_proof_shplonk, /root/reference/src/bin/cli.rs:519]:
// a byte buffer; a point appends its canonical x and y as 32 BIG-endian bytes each, a scalar its 32 big-endian bytes; a challenge is
// Keccak-256 of the buffer (plus one 0x01 byte when the buffer is exactly 32 bytes, i.e. two squeezes in a row), read as a
// big-endian integer mod r, and the digest becomes the new buffer.  The proof stream receives the same 64 / 32 bytes.
namespace {
struct Keccak256 {
    static void f1600(uint64_t st[25]) {
        static const uint64_t RC[24] = {0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
                                        0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
                                        0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
                                        0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
                                        0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
                                        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
        static const int ROT[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
        static const int PIL[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
        auto rotl = [](uint64_t x, int n) { return (x << n) | (x >> (64 - n)); };
        for (int r = 0; r < 24; ++r) {
            uint64_t bc[5];
            for (int i = 0; i < 5; ++i) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
            for (int i = 0; i < 5; ++i) { uint64_t t = bc[(i + 4) % 5] ^ rotl(bc[(i + 1) % 5], 1); for (int j = 0; j < 25; j += 5) st[j + i] ^= t; }
            uint64_t t = st[1];
            for (int i = 0; i < 24; ++i) { int j = PIL[i]; uint64_t b = st[j]; st[j] = rotl(t, ROT[i]); t = b; }
            for (int j = 0; j < 25; j += 5) {
                for (int i = 0; i < 5; ++i) bc[i] = st[j + i];
                for (int i = 0; i < 5; ++i) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
            }
            st[0] ^= RC[r];
        }
    }
    // pad = 0x01 for Keccak-256 (Ethereum), 0x06 for SHA3-256 (used by the self-check against hashlib)
    static void digest(const uint8_t* in, size_t len, uint8_t pad, uint8_t out[32]) {
        const size_t rate = 136;
        uint64_t st[25] = {0};
        uint8_t block[136];
        while (len >= rate) {
            for (size_t i = 0; i < rate / 8; ++i) { uint64_t w; memcpy(&w, in + 8 * i, 8); st[i] ^= w; }
            f1600(st);
            in += rate; len -= rate;
        }
        memset(block, 0, rate);
        memcpy(block, in, len);
        block[len] ^= pad;
        block[rate - 1] ^= 0x80;
        for (size_t i = 0; i < rate / 8; ++i) { uint64_t w; memcpy(&w, block + 8 * i, 8); st[i] ^= w; }
        f1600(st);
        memcpy(out, st, 32);
    }
};
}  // namespace

struct zkhip_evm_transcript {
    zk_transcript cb;
    std::vector<uint8_t> buf, proof;
    std::vector<uint64_t> points_xy, challenges;
};

namespace {
inline void be32(const fe32& canon, uint8_t out[32]) {
    const uint8_t* le = (const uint8_t*)canon.w;
    for (int i = 0; i < 32; ++i) out[i] = le[31 - i];
}
void e_write_point(void* user, const uint8_t*, const uint64_t xy[8]) {
    auto* t = (zkhip_evm_transcript*)user;
    uint8_t b[64];
    be32(abi_to_canonical_words<Fq>(mem_load(xy)), b);
    be32(abi_to_canonical_words<Fq>(mem_load(xy + 4)), b + 32);
    t->buf.insert(t->buf.end(), b, b + 64);
    t->proof.insert(t->proof.end(), b, b + 64);
    t->points_xy.insert(t->points_xy.end(), xy, xy + 8);
}
void e_write_scalar(void* user, const uint64_t scalar[4]) {
    auto* t = (zkhip_evm_transcript*)user;
    uint8_t b[32];
    be32(abi_to_canonical_words<Fr>(mem_load(scalar)), b);
    t->buf.insert(t->buf.end(), b, b + 32);
    t->proof.insert(t->proof.end(), b, b + 32);
}
void e_squeeze(void* user, uint64_t out[4]) {
    auto* t = (zkhip_evm_transcript*)user;
    if (t->buf.size() == 32) t->buf.push_back(1);
    uint8_t d[32];
    Keccak256::digest(t->buf.data(), t->buf.size(), 0x01, d);
    t->buf.assign(d, d + 32);
    uint32_t w[8];   // the digest as a big-endian integer -> little-endian words -> mod r (from_canonical_words reduces)
    for (int i = 0; i < 8; ++i) w[i] = ((uint32_t)d[31 - 4 * i]) | ((uint32_t)d[30 - 4 * i] << 8) | ((uint32_t)d[29 - 4 * i] << 16) | ((uint32_t)d[28 - 4 * i] << 24);
    fe32 abi = hf_abi(hf_from_canonical_words(w));
    memcpy(out, abi.w, 32);
    t->challenges.insert(t->challenges.end(), out, out + 4);
}
}  // namespace

extern "C" {

zkhip_evm_transcript* zkhip_evm_transcript_new(void) {
    auto* t = new zkhip_evm_transcript();
    t->cb.user = t;
    t->cb.write_point = e_write_point;
    t->cb.squeeze_challenge = e_squeeze;
    t->cb.write_scalar = e_write_scalar;
    return t;
}
void zkhip_evm_transcript_free(zkhip_evm_transcript* t) { delete t; }
const zk_transcript* zkhip_evm_transcript_callbacks(zkhip_evm_transcript* t) { return t ? &t->cb : nullptr; }
size_t zkhip_evm_transcript_proof(const zkhip_evm_transcript* t, const uint8_t** bytes) {
    if (!t) return 0;
    if (bytes) *bytes = t->proof.data();
    return t->proof.size();
}
size_t zkhip_evm_transcript_challenges(const zkhip_evm_transcript* t, const uint64_t** limbs) {
    if (!t) return 0;
    if (limbs) *limbs = t->challenges.data();
    return t->challenges.size() / 4;
}
/* Keccak-f[1600] sponge with a caller-chosen padding byte (0x01 Keccak-256, 0x06 SHA3-256): exposed so that the permutation can be
 * checked against a library implementation of SHA3-256 */
void zkhip_keccak256(const uint8_t* in, size_t len, uint8_t pad, uint8_t out[32]) { Keccak256::digest(in, len, pad, out); }

zkhip_blake2b_transcript* zkhip_blake2b_transcript_new(void) {
    auto* t = new zkhip_blake2b_transcript();
    t->state.init("Halo2-Transcript");
    t->cb.user = t;
    t->cb.write_point = t_write_point;
    t->cb.squeeze_challenge = t_squeeze;
    t->cb.write_scalar = t_write_scalar;
    return t;
}
void zkhip_blake2b_transcript_free(zkhip_blake2b_transcript* t) { delete t; }
const zk_transcript* zkhip_blake2b_transcript_callbacks(zkhip_blake2b_transcript* t) { return t ? &t->cb : nullptr; }
size_t zkhip_blake2b_transcript_proof(const zkhip_blake2b_transcript* t, const uint8_t** bytes) {
    if (!t) return 0;
    if (bytes) *bytes = t->proof.data();
    return t->proof.size();
}
size_t zkhip_blake2b_transcript_points(const zkhip_blake2b_transcript* t, const uint64_t** xy) {
    if (!t) return 0;
    if (xy) *xy = t->points_xy.data();
    return t->points_xy.size() / 8;
}
size_t zkhip_blake2b_transcript_challenges(const zkhip_blake2b_transcript* t, const uint64_t** limbs) {
    if (!t) return 0;
    if (limbs) *limbs = t->challenges.data();
    return t->challenges.size() / 4;
}

}  // extern "C"
