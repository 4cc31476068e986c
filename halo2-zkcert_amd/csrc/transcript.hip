// transcript.hip — halo2_proofs transcript.rs Blake2bWrite<Vec<u8>, G1Affine, Challenge255<G1Affine>> on the host
// [UPSTREAM-RECALL; crate pinned at /root/reference/Cargo.lock:1320-1322], as a ready-made zk_transcript for zkhip_create_proof:
// BLAKE2b-512 personalised "Halo2-Transcript"; write_point absorbs prefix 1 and the canonical x, y (32 little-endian bytes each)
// and appends the 32-byte compressed point to the proof; write_scalar absorbs prefix 2 and the canonical scalar and appends
// its 32 bytes; a challenge absorbs prefix 0 and is the 64-byte digest of a copy of the state, reduced mod r (little-endian).
// Further down: snark-verifier's Keccak EvmTranscript and its PoseidonTranscript — the two transcripts the reference's own commands use
// (gen_evm_proof_shplonk / gen_snark_shplonk).  A Rust caller passes callbacks into its own transcript instead.
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
void t_common_scalar(void* user, const uint64_t scalar[4]) {   // Transcript::common_scalar: absorbed like a written scalar, no proof bytes
    auto* t = (zkhip_blake2b_transcript*)user;
    uint8_t msg[33];
    msg[0] = 2;
    fe32 s = abi_to_canonical_words<Fr>(mem_load(scalar));
    memcpy(msg + 1, s.w, 32);
    t->state.update(msg, 33);
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
void e_common_scalar(void* user, const uint64_t scalar[4]) {
    auto* t = (zkhip_evm_transcript*)user;
    uint8_t b[32];
    be32(abi_to_canonical_words<Fr>(mem_load(scalar)), b);
    t->buf.insert(t->buf.end(), b, b + 32);
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

// ------------------------------------------------------------------ Poseidon transcript
// snark-verifier system/halo2/transcript/halo2.rs PoseidonTranscript<G1Affine, NativeLoader, Vec<u8>, T = 3, RATE = 2, R_F = 8,
// R_P = 57> over util/hash/poseidon.rs [UPSTREAM-RECALL; crate pinned at /root/reference/Cargo.lock:2675-2693 / 2714-2716; the
// transcript behind gen_snark_shplonk, /root/reference/src/helpers.rs:233,299 and src/bin/cli.rs:320,343,369,462]:
//  * the permutation is Poseidon x^5 over BN254 Fr, t = 3, 8 full + 57 partial rounds, round constants and the Cauchy MDS matrix from
//    the Grain LFSR of the Poseidon reference (field type 1, s-box 0, 254 bits; constants by rejection sampling, the 2t MDS seeds
//    reduced mod r; first MDS candidate = SECURE_MDS 0).  The generator below reproduces the published permutation vector
//    poseidonperm_x5_254_3 (tests/golden/poseidon.json) — the permutation is pinned against an external source.  Upstream runs
//    the "optimized" form (pre-sparse MDS, shifted constants), which computes the same permutation;
//  * sponge (from recall, not externally pinned): state = [2^64, 0, 0]; absorbed elements are buffered; squeeze() permutes once
//    per chunk of RATE elements (state[1 + i] += chunk[i]; a short chunk adds 1 at position len + 1), once more on an empty
//    chunk when the buffer length is a multiple of RATE (including 0), and returns state[1];
//  * a point is absorbed as its x and y coordinates reduced into Fr (fe_to_fe), a scalar as itself; the proof stream receives the
//    32-byte compressed point / the 32-byte little-endian scalar.
namespace {
struct PoseidonSpec {
    static constexpr int T = 3, RATE = 2, R_F = 8, R_P = 57;
    HF rc[R_F + R_P][T];
    HF mds[T][T];
    struct Grain {
        uint8_t st[80];
        int pos = 0;   // ring buffer start
        int bit() {
            auto at = [&](int i) { return st[(pos + i) % 80]; };
            uint8_t b = at(62) ^ at(51) ^ at(38) ^ at(23) ^ at(13) ^ at(0);
            st[pos] = b;
            pos = (pos + 1) % 80;
            return b;
        }
        int filtered() {
            for (;;) { int a = bit(), b = bit(); if (a) return b; }
        }
        void raw(uint32_t w[8]) {   // 254 bits, most significant first
            memset(w, 0, 32);
            for (int i = 253; i >= 0; --i) if (filtered()) w[i >> 5] |= 1u << (i & 31);
        }
    };
    static bool less_than_r(const uint32_t w[8]) {
        for (int i = 3; i >= 0; --i) {
            uint64_t v = w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
            if (v != hostfr::P[i]) return v < hostfr::P[i];
        }
        return false;
    }
    PoseidonSpec() {
        Grain g;
        int n = 0;
        auto push = [&](uint32_t v, int bits) { for (int i = bits - 1; i >= 0; --i) g.st[n++] = (v >> i) & 1; };
        push(1, 2); push(0, 4); push(254, 12); push(T, 12); push(R_F, 10); push(R_P, 10);
        for (int i = 0; i < 30; ++i) g.st[n++] = 1;
        for (int i = 0; i < 160; ++i) g.bit();
        uint32_t w[8];
        for (int r = 0; r < R_F + R_P; ++r)
            for (int i = 0; i < T; ++i) {
                do g.raw(w); while (!less_than_r(w));
                rc[r][i] = hf_from_canonical_words(w);
            }
        HF xs[T], ys[T];
        for (int i = 0; i < T; ++i) { g.raw(w); xs[i] = hf_from_canonical_words(w); }   // reduced mod r, no rejection
        for (int i = 0; i < T; ++i) { g.raw(w); ys[i] = hf_from_canonical_words(w); }
        for (int i = 0; i < T; ++i)
            for (int j = 0; j < T; ++j) mds[i][j] = hf_invert(hadd(xs[i], ys[j]));
        build_sparse();
    }
    // ---- the partial rounds in sparse form (the optimisation of the Poseidon paper's appendix, derived here directly):
    // a partial round is x -> M S(x + c) with S the s-box on coordinate 0 only.  Write the running state as x_j = B_j z_j with
    // B_j = diag(1, B^_j) (B_0 = I): then x_(j+1) = (M B_j) S(z_j + B_j^-1 c_j) because diag(1, .) commutes with S.  Factor
    // A_j = M B_j = A'_j A''_j with A'_j = diag(1, A^_j) (A^_j = the lower-right block of A_j) and the SPARSE
    // A''_j = [[A_00, v], [A^_j^-1 w, I]] (v = first row, w = first column of A_j below / right of A_00); z_(j+1) = A''_j S(z_j + c'_j),
    // B_(j+1) = A'_j.  After the last partial round the dense B_RP is applied once.  Per partial round: one 3-term dot product and
    // two multiply-adds instead of a dense 3 x 3 product.  permute() == permute_plain() is checked on the published vector and on
    // random states (tests/test_schedule_cpu.py through zkhip_poseidon_permute).
    HF sp_c[R_P][T];        // transformed constants c'_j
    HF sp_row[R_P][T];      // (A_00, v_1, v_2)
    HF sp_col[R_P][T - 1];  // A^_j^-1 w
    HF sp_last[T - 1][T - 1];   // B^_RP
    static void inv2(const HF m[2][2], HF o[2][2]) {
        const HF det = hsub(hmul(m[0][0], m[1][1]), hmul(m[0][1], m[1][0]));
        const HF di = hf_invert(det);
        o[0][0] = hmul(m[1][1], di);
        o[1][1] = hmul(m[0][0], di);
        o[0][1] = hmul(hsub(hzero(), m[0][1]), di);
        o[1][0] = hmul(hsub(hzero(), m[1][0]), di);
    }
    void build_sparse() {
        static_assert(T == 3, "the sparse form below is written for t = 3");
        HF B[2][2] = {{hone(), hzero()}, {hzero(), hone()}};
        for (int j = 0; j < R_P; ++j) {
            const HF* c = rc[R_F / 2 + j];
            HF Bi[2][2];
            inv2(B, Bi);
            sp_c[j][0] = c[0];
            for (int i = 0; i < 2; ++i) sp_c[j][1 + i] = hadd(hmul(Bi[i][0], c[1]), hmul(Bi[i][1], c[2]));
            // A = M diag(1, B): first column = M's; A[r][1 + q] = sum_k M[r][1 + k] B[k][q]
            HF A[3][3];
            for (int r = 0; r < 3; ++r) {
                A[r][0] = mds[r][0];
                for (int q = 0; q < 2; ++q) A[r][1 + q] = hadd(hmul(mds[r][1], B[0][q]), hmul(mds[r][2], B[1][q]));
            }
            HF Ah[2][2] = {{A[1][1], A[1][2]}, {A[2][1], A[2][2]}}, Ahi[2][2];
            inv2(Ah, Ahi);
            sp_row[j][0] = A[0][0]; sp_row[j][1] = A[0][1]; sp_row[j][2] = A[0][2];
            for (int i = 0; i < 2; ++i) sp_col[j][i] = hadd(hmul(Ahi[i][0], A[1][0]), hmul(Ahi[i][1], A[2][0]));
            for (int i = 0; i < 2; ++i) for (int q = 0; q < 2; ++q) B[i][q] = Ah[i][q];
        }
        for (int i = 0; i < 2; ++i) for (int q = 0; q < 2; ++q) sp_last[i][q] = B[i][q];
    }
    static const PoseidonSpec& get() { static const PoseidonSpec s; return s; }
    static inline HF pow5(const HF& a) { HF a2 = hmul(a, a); return hmul(hmul(a2, a2), a); }
    void mix(HF s[T]) const {
        HF o[T];
        for (int i = 0; i < T; ++i) o[i] = hdot3(mds[i][0], s[0], mds[i][1], s[1], mds[i][2], s[2]);
        for (int i = 0; i < T; ++i) s[i] = o[i];
    }
    void permute_plain(HF s[T]) const {   // the textbook form: every round a dense MDS product
        int r = 0;
        for (int f = 0; f < R_F / 2; ++f, ++r) { for (int i = 0; i < T; ++i) s[i] = pow5(hadd(s[i], rc[r][i])); mix(s); }
        for (int p_ = 0; p_ < R_P; ++p_, ++r) { for (int i = 0; i < T; ++i) s[i] = hadd(s[i], rc[r][i]); s[0] = pow5(s[0]); mix(s); }
        for (int f = 0; f < R_F / 2; ++f, ++r) { for (int i = 0; i < T; ++i) s[i] = pow5(hadd(s[i], rc[r][i])); mix(s); }
    }
    void permute(HF s[T]) const {
        int r = 0;
        for (int f = 0; f < R_F / 2; ++f, ++r) { for (int i = 0; i < T; ++i) s[i] = pow5(hadd(s[i], rc[r][i])); mix(s); }
        for (int j = 0; j < R_P; ++j, ++r) {
            const HF x0 = pow5(hadd(s[0], sp_c[j][0])), x1 = hadd(s[1], sp_c[j][1]), x2 = hadd(s[2], sp_c[j][2]);
            s[0] = hdot3(sp_row[j][0], x0, sp_row[j][1], x1, sp_row[j][2], x2);
            s[1] = hadd(hmul(sp_col[j][0], x0), x1);
            s[2] = hadd(hmul(sp_col[j][1], x0), x2);
        }
        {
            const HF y1 = hadd(hmul(sp_last[0][0], s[1]), hmul(sp_last[0][1], s[2])), y2 = hadd(hmul(sp_last[1][0], s[1]), hmul(sp_last[1][1], s[2]));
            s[1] = y1; s[2] = y2;
        }
        for (int f = 0; f < R_F / 2; ++f, ++r) { for (int i = 0; i < T; ++i) s[i] = pow5(hadd(s[i], rc[r][i])); mix(s); }
    }
};
}  // namespace

struct zkhip_poseidon_transcript {
    zk_transcript cb;
    HF state[3];
    // upstream buffers everything absorbed since the last squeeze and permutes chunk by chunk inside squeeze(); the same chunks in
    // the same order are permuted here as soon as they are complete, so that elements absorbed while the GPU is busy (the
    // verifying key, the instance values) cost nothing at the next Fiat-Shamir point.  `pending` holds < RATE elements.
    HF pending[2];
    size_t n_pending = 0;
    void absorb(const HF& e);
    std::vector<uint8_t> proof;
    std::vector<uint64_t> points_xy, challenges;
};

void zkhip_poseidon_transcript::absorb(const HF& e) {
    pending[n_pending++] = e;
    if (n_pending == (size_t)PoseidonSpec::RATE) {   // a full chunk: state[1 + i] += chunk[i], permute
        for (int i = 0; i < PoseidonSpec::RATE; ++i) state[1 + i] = hadd(state[1 + i], pending[i]);
        n_pending = 0;
        PoseidonSpec::get().permute(state);
    }
}

namespace {
inline HF fq_coordinate_in_fr(const uint64_t* mont_fq) {   // fe_to_fe: the integer value of an Fq element, reduced mod r
    fe32 c = abi_to_canonical_words<Fq>(mem_load(mont_fq));
    return hf_from_canonical_words(c.w);
}
void p_absorb_point(zkhip_poseidon_transcript* t, const uint64_t xy[8]) {
    t->absorb(fq_coordinate_in_fr(xy));
    t->absorb(fq_coordinate_in_fr(xy + 4));
}
void p_write_point(void* user, const uint8_t bytes32[32], const uint64_t xy[8]) {
    auto* t = (zkhip_poseidon_transcript*)user;
    p_absorb_point(t, xy);
    t->proof.insert(t->proof.end(), bytes32, bytes32 + 32);
    t->points_xy.insert(t->points_xy.end(), xy, xy + 8);
}
void p_common_scalar(void* user, const uint64_t scalar[4]) {
    auto* t = (zkhip_poseidon_transcript*)user;
    t->absorb(hf_from_abi(scalar));
}
void p_write_scalar(void* user, const uint64_t scalar[4]) {
    auto* t = (zkhip_poseidon_transcript*)user;
    t->absorb(hf_from_abi(scalar));
    fe32 s = abi_to_canonical_words<Fr>(mem_load(scalar));
    const uint8_t* b = (const uint8_t*)s.w;
    t->proof.insert(t->proof.end(), b, b + 32);
}
void p_squeeze(void* user, uint64_t out[4]) {
    auto* t = (zkhip_poseidon_transcript*)user;
    const PoseidonSpec& sp = PoseidonSpec::get();
    // the last chunk: the pending elements (a short chunk adds 1 at position len + 1); an empty one exactly when the absorbed length
    // was a multiple of RATE (upstream's `exact`)
    for (size_t i = 0; i < t->n_pending; ++i) t->state[1 + i] = hadd(t->state[1 + i], t->pending[i]);
    t->state[t->n_pending + 1] = hadd(t->state[t->n_pending + 1], hone());
    t->n_pending = 0;
    sp.permute(t->state);
    fe32 abi = hf_abi(t->state[1]);
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
    t->cb.common_scalar = e_common_scalar;
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
    t->cb.common_scalar = t_common_scalar;
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

zkhip_poseidon_transcript* zkhip_poseidon_transcript_new(void) {
    auto* t = new zkhip_poseidon_transcript();
    t->state[0] = HF{{0, 1, 0, 0}};                       // 2^64 as an integer ...
    t->state[0] = hmul(t->state[0], hf_from_abi(HF{{0x1bb8e645ae216da7ull, 0x53fe3ab1e35c59e3ull, 0x8c49833d53bb8085ull, 0x0216d0b17f4e44a5ull}}.w));   // ... into Montgomery form (x R^2 / R)
    t->state[1] = t->state[2] = hzero();
    t->n_pending = 0;
    t->cb.user = t;
    t->cb.write_point = p_write_point;
    t->cb.squeeze_challenge = p_squeeze;
    t->cb.write_scalar = p_write_scalar;
    t->cb.common_scalar = p_common_scalar;
    return t;
}
void zkhip_poseidon_transcript_free(zkhip_poseidon_transcript* t) { delete t; }
const zk_transcript* zkhip_poseidon_transcript_callbacks(zkhip_poseidon_transcript* t) { return t ? &t->cb : nullptr; }
size_t zkhip_poseidon_transcript_proof(const zkhip_poseidon_transcript* t, const uint8_t** bytes) {
    if (!t) return 0;
    if (bytes) *bytes = t->proof.data();
    return t->proof.size();
}
size_t zkhip_poseidon_transcript_points(const zkhip_poseidon_transcript* t, const uint64_t** xy) {
    if (!t) return 0;
    if (xy) *xy = t->points_xy.data();
    return t->points_xy.size() / 8;
}
size_t zkhip_poseidon_transcript_challenges(const zkhip_poseidon_transcript* t, const uint64_t** limbs) {
    if (!t) return 0;
    if (limbs) *limbs = t->challenges.data();
    return t->challenges.size() / 4;
}
/* The bare permutation (state: 3 x 4 u64, ABI form, in place) and the generated parameters (rc: 65 x 3, mds: 3 x 3 row-major; ABI
 * form): exposed so that the generator can be checked against the published poseidonperm_x5_254_3 vector. */
void zkhip_poseidon_permute_plain(uint64_t state[12]) {   // textbook rounds (dense MDS everywhere): the cross-check of the sparse form
    HF s[3];
    for (int i = 0; i < 3; ++i) s[i] = hf_from_abi(state + 4 * i);
    PoseidonSpec::get().permute_plain(s);
    for (int i = 0; i < 3; ++i) { fe32 o = hf_abi(s[i]); memcpy(state + 4 * i, o.w, 32); }
}
void zkhip_poseidon_permute(uint64_t state[12]) {
    HF s[3];
    for (int i = 0; i < 3; ++i) s[i] = hf_from_abi(state + 4 * i);
    PoseidonSpec::get().permute(s);
    for (int i = 0; i < 3; ++i) { fe32 o = hf_abi(s[i]); memcpy(state + 4 * i, o.w, 32); }
}
void zkhip_poseidon_params(uint64_t* rc, uint64_t* mds) {
    const PoseidonSpec& sp = PoseidonSpec::get();
    if (rc) for (int r = 0; r < 65; ++r) for (int i = 0; i < 3; ++i) { fe32 o = hf_abi(sp.rc[r][i]); memcpy(rc + 4 * (3 * r + i), o.w, 32); }
    if (mds) for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { fe32 o = hf_abi(sp.mds[i][j]); memcpy(mds + 4 * (3 * i + j), o.w, 32); }
}
size_t zkhip_evm_transcript_points(const zkhip_evm_transcript* t, const uint64_t** xy) {
    if (!t) return 0;
    if (xy) *xy = t->points_xy.data();
    return t->points_xy.size() / 8;
}

}  // extern "C"
