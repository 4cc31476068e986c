#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <chrono>
typedef unsigned __int128 u128;
static const uint64_t P[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static const uint64_t INV = 0xc2e1f593efffffffull;
inline bool geq(const uint64_t* a, const uint64_t* P) { for (int i = 3; i >= 0; --i) if (a[i] != P[i]) return a[i] > P[i]; return true; }
inline void sub_mod(uint64_t* a, const uint64_t* P) { u128 br = 0; for (int i = 0; i < 4; ++i) { u128 d = (u128)a[i] - P[i] - (uint64_t)br; a[i] = (uint64_t)d; br = (d >> 64) & 1; } }
inline void mul_old(uint64_t* r, const uint64_t* a, const uint64_t* b) {
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
// no-carry CIOS (the modulus' top limb leaves two spare bits): one carry chain per half-iteration, no t[4]/t[5]
inline void mul_new(uint64_t* r, const uint64_t* a, const uint64_t* b) {
    uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint64_t bi = b[i];
        u128 A = (u128)a[0] * bi + t0;
        uint64_t lo = (uint64_t)A;
        const uint64_t m = lo * INV;
        u128 C = ((u128)m * P[0] + lo) >> 64;
        A = (A >> 64) + (u128)a[1] * bi + t1;
        C += (u128)m * P[1] + (uint64_t)A; t0 = (uint64_t)C; C >>= 64;
        A = (A >> 64) + (u128)a[2] * bi + t2;
        C += (u128)m * P[2] + (uint64_t)A; t1 = (uint64_t)C; C >>= 64;
        A = (A >> 64) + (u128)a[3] * bi + t3;
        C += (u128)m * P[3] + (uint64_t)A; t2 = (uint64_t)C; C >>= 64;
        t3 = (uint64_t)C + (uint64_t)(A >> 64);
    }
    r[0] = t0; r[1] = t1; r[2] = t2; r[3] = t3;
    if (geq(r, P)) sub_mod(r, P);
}
int main() {
    uint64_t a[4] = {0x1234567890abcdefull, 0xfedcba0987654321ull, 0x0f0f0f0f0f0f0f0full, 0x1111111111111111ull}, b[4] = {5, 6, 7, 0x2000000000000000ull};
    uint64_t x[4], y[4];
    memcpy(x, a, 32); memcpy(y, a, 32);
    for (int i = 0; i < 1000; ++i) { mul_old(x, x, b); mul_new(y, y, b); if (memcmp(x, y, 32)) { printf("MISMATCH at %d\n", i); return 1; } mul_old(b, b, x); }
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 20000000; ++i) mul_old(x, x, b);
    auto t1 = std::chrono::steady_clock::now();
    for (int i = 0; i < 20000000; ++i) mul_new(y, y, b);
    auto t2 = std::chrono::steady_clock::now();
    printf("old %.2f ns new %.2f ns (%llx %llx)\n", std::chrono::duration<double, std::nano>(t1 - t0).count() / 2e7, std::chrono::duration<double, std::nano>(t2 - t1).count() / 2e7, (unsigned long long)x[0], (unsigned long long)y[0]);
}
