/* zkoracle.c — CPU restatement of the create_proof hot path.  TEST INFRASTRUCTURE ONLY.
 * parity unpinned — see zkoracle.h for the full statement and the reference call sites.
 *
 * Each block names the upstream file it restates ([UPSTREAM-RECALL]: the crates are not on
 * this machine; pins at /root/reference/Cargo.lock:1320-1322 and :1359-1361).
 */
#include "zkoracle.h"
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fe;
typedef struct { uint64_t m[4]; uint64_t inv; fe one; fe r2; } fparams;

/* ------------------------------------------------------------------ a1: field arithmetic
 * halo2curves src/bn256/fq.rs, fr.rs (constants), src/derive/field.rs (4-limb Montgomery). */
static const fparams FQ = {
    {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
    0x87d20782e4866389ULL,
    {{0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL}},
    {{0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL}}};
static const fparams FR = {
    {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL},
    0xc2e1f593efffffffULL,
    {{0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL}},
    {{0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL}}};

static inline int fe_is_zero(const fe* a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int fe_eq(const fe* a, const fe* b) {
    return ((a->l[0] ^ b->l[0]) | (a->l[1] ^ b->l[1]) | (a->l[2] ^ b->l[2]) | (a->l[3] ^ b->l[3])) == 0;
}
static inline int geq_mod(const uint64_t t[4], const uint64_t m[4]) {
    for (int i = 3; i >= 0; --i) {
        if (t[i] > m[i]) return 1;
        if (t[i] < m[i]) return 0;
    }
    return 1;
}
static inline void sub_mod_raw(uint64_t t[4], const uint64_t m[4]) {
    u128 b = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)t[i] - m[i] - (uint64_t)b;
        t[i] = (uint64_t)d;
        b = (d >> 64) & 1;
    }
}
static inline void f_add(const fparams* F, const fe* a, const fe* b, fe* o) {
    u128 c = 0;
    uint64_t t[4];
    for (int i = 0; i < 4; ++i) {
        c += (u128)a->l[i] + b->l[i];
        t[i] = (uint64_t)c;
        c >>= 64;
    }
    /* both moduli are 254-bit: no carry out of 256 bits */
    if (geq_mod(t, F->m)) sub_mod_raw(t, F->m);
    memcpy(o->l, t, 32);
}
static inline void f_sub(const fparams* F, const fe* a, const fe* b, fe* o) {
    u128 br = 0;
    uint64_t t[4];
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a->l[i] - b->l[i] - (uint64_t)br;
        t[i] = (uint64_t)d;
        br = (d >> 64) & 1;
    }
    if (br) {
        u128 c = 0;
        for (int i = 0; i < 4; ++i) {
            c += (u128)t[i] + F->m[i];
            t[i] = (uint64_t)c;
            c >>= 64;
        }
    }
    memcpy(o->l, t, 32);
}
static inline void f_neg(const fparams* F, const fe* a, fe* o) {
    fe z = {{0, 0, 0, 0}};
    f_sub(F, &z, a, o);
}
static inline void f_dbl(const fparams* F, const fe* a, fe* o) { f_add(F, a, a, o); }

/* CIOS Montgomery product a*b*2^-256 mod m, fully reduced. */
static inline void f_mul(const fparams* F, const fe* a, const fe* b, fe* o) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        u128 c = 0;
        for (int j = 0; j < 4; ++j) {
            c += (u128)a->l[j] * b->l[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (uint64_t)c;
        t[5] = (uint64_t)(c >> 64);
        uint64_t q = t[0] * F->inv;
        c = (u128)q * F->m[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 4; ++j) {
            c += (u128)q * F->m[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (uint64_t)c;
        t[4] = t[5] + (uint64_t)(c >> 64);
    }
    if (t[4] || geq_mod(t, F->m)) sub_mod_raw(t, F->m);
    memcpy(o->l, t, 32);
}
static inline void f_sqr(const fparams* F, const fe* a, fe* o) { f_mul(F, a, a, o); }

static void f_pow(const fparams* F, const fe* a, const uint64_t e[4], fe* o) {
    fe acc = F->one, base = *a;
    int started = 0;
    for (int i = 255; i >= 0; --i) {
        if (started) f_sqr(F, &acc, &acc);
        if ((e[i / 64] >> (i % 64)) & 1) {
            if (started) f_mul(F, &acc, &base, &acc);
            else { acc = base; started = 1; }
        }
    }
    if (!started) acc = F->one;
    *o = acc;
}
static void f_inv(const fparams* F, const fe* a, fe* o) {
    if (fe_is_zero(a)) { memset(o, 0, sizeof *o); return; }
    uint64_t e[4];
    memcpy(e, F->m, 32);
    e[0] -= 2; /* m[0] >= 2 for both moduli */
    f_pow(F, a, e, o);
}
static inline void f_from_mont(const fparams* F, const fe* a, fe* o) {
    fe one = {{1, 0, 0, 0}};
    f_mul(F, a, &one, o);
}
static inline void f_to_mont(const fparams* F, const fe* a, fe* o) { f_mul(F, a, &F->r2, o); }
static inline void f_from_u64(const fparams* F, uint64_t v, fe* o) {
    fe t = {{v, 0, 0, 0}};
    f_to_mont(F, &t, o);
}

#define FE(p) ((const fe*)(p))
#define FEM(p) ((fe*)(p))
void zko_fr_mul(const uint64_t a[4], const uint64_t b[4], uint64_t o[4]) { f_mul(&FR, FE(a), FE(b), FEM(o)); }
void zko_fr_add(const uint64_t a[4], const uint64_t b[4], uint64_t o[4]) { f_add(&FR, FE(a), FE(b), FEM(o)); }
void zko_fr_sub(const uint64_t a[4], const uint64_t b[4], uint64_t o[4]) { f_sub(&FR, FE(a), FE(b), FEM(o)); }
void zko_fr_inv(const uint64_t a[4], uint64_t o[4]) { f_inv(&FR, FE(a), FEM(o)); }
void zko_fr_pow(const uint64_t a[4], const uint64_t e[4], uint64_t o[4]) { f_pow(&FR, FE(a), e, FEM(o)); }
void zko_fr_to_repr(const uint64_t a[4], uint64_t o[4]) { f_from_mont(&FR, FE(a), FEM(o)); }
void zko_fr_from_repr(const uint64_t a[4], uint64_t o[4]) { f_to_mont(&FR, FE(a), FEM(o)); }
void zko_fq_mul(const uint64_t a[4], const uint64_t b[4], uint64_t o[4]) { f_mul(&FQ, FE(a), FE(b), FEM(o)); }
void zko_fq_add(const uint64_t a[4], const uint64_t b[4], uint64_t o[4]) { f_add(&FQ, FE(a), FE(b), FEM(o)); }
void zko_fq_sub(const uint64_t a[4], const uint64_t b[4], uint64_t o[4]) { f_sub(&FQ, FE(a), FE(b), FEM(o)); }
void zko_fq_inv(const uint64_t a[4], uint64_t o[4]) { f_inv(&FQ, FE(a), FEM(o)); }
void zko_fq_to_repr(const uint64_t a[4], uint64_t o[4]) { f_from_mont(&FQ, FE(a), FEM(o)); }
void zko_fq_from_repr(const uint64_t a[4], uint64_t o[4]) { f_to_mont(&FQ, FE(a), FEM(o)); }

/* Fr::ROOT_OF_UNITY (2^28-th), Fr::ZETA, Fr::DELTA — canonical values, halo2curves fr.rs */
static const fe FR_ROOT_CANON = {{0xd34f1ed960c37c9cULL, 0x3215cf6dd39329c8ULL, 0x98865ea93dd31f74ULL, 0x03ddb9f5166d18b7ULL}};
static const fe FR_ZETA_CANON = {{0x8b17ea66b99c90ddULL, 0x5bfc41088d8daaa7ULL, 0xb3c4d79d41a91758ULL, 0}};
static const fe FR_DELTA_CANON = {{0x870e56bbe533e9a2ULL, 0x5b5f898e5e963f25ULL, 0x64ec26aad4c86e71ULL, 0x09226b6e22c6f0caULL}};
#define FR_S 28

void zko_fr_root_of_unity(uint32_t k, uint64_t out[4]) {
    fe w;
    f_to_mont(&FR, &FR_ROOT_CANON, &w);
    for (uint32_t i = k; i < FR_S; ++i) f_sqr(&FR, &w, &w);
    memcpy(out, w.l, 32);
}
void zko_fr_constants(uint64_t zeta[4], uint64_t delta[4]) {
    f_to_mont(&FR, &FR_ZETA_CANON, FEM(zeta));
    f_to_mont(&FR, &FR_DELTA_CANON, FEM(delta));
}

/* ------------------------------------------------------------------ a2: G1
 * halo2curves src/derive/curve.rs new_curve_impl!: Jacobian {x,y,z}, a = 0, b = 3. */
typedef struct { fe x, y; } g1a;       /* identity: x = y = 0 */
typedef struct { fe x, y, z; } g1j;    /* identity: z = 0 */

static inline int g1a_is_id(const g1a* p) { return fe_is_zero(&p->x) && fe_is_zero(&p->y); }
static inline int g1j_is_id(const g1j* p) { return fe_is_zero(&p->z); }
static inline void g1j_set_id(g1j* p) { memset(p, 0, sizeof *p); p->y = FQ.one; }
static inline void g1j_from_affine(const g1a* a, g1j* o) {
    if (g1a_is_id(a)) { g1j_set_id(o); return; }
    o->x = a->x; o->y = a->y; o->z = FQ.one;
}
/* dbl-2009-l */
static void g1j_double(const g1j* p, g1j* o) {
    if (g1j_is_id(p)) { g1j_set_id(o); return; }
    fe a, b, c, d, e, f, t, x3, y3, z3;
    f_sqr(&FQ, &p->x, &a);
    f_sqr(&FQ, &p->y, &b);
    f_sqr(&FQ, &b, &c);
    f_add(&FQ, &p->x, &b, &d); f_sqr(&FQ, &d, &d); f_sub(&FQ, &d, &a, &d); f_sub(&FQ, &d, &c, &d); f_dbl(&FQ, &d, &d);
    f_dbl(&FQ, &a, &e); f_add(&FQ, &e, &a, &e);
    f_sqr(&FQ, &e, &f);
    f_mul(&FQ, &p->z, &p->y, &z3); f_dbl(&FQ, &z3, &z3);
    f_dbl(&FQ, &d, &t); f_sub(&FQ, &f, &t, &x3);
    f_dbl(&FQ, &c, &c); f_dbl(&FQ, &c, &c); f_dbl(&FQ, &c, &c);
    f_sub(&FQ, &d, &x3, &t); f_mul(&FQ, &e, &t, &y3); f_sub(&FQ, &y3, &c, &y3);
    o->x = x3; o->y = y3; o->z = z3;
}
/* add-2007-bl with the exceptional cases */
static void g1j_add(const g1j* p, const g1j* q, g1j* o) {
    if (g1j_is_id(p)) { *o = *q; return; }
    if (g1j_is_id(q)) { *o = *p; return; }
    fe z1z1, z2z2, u1, u2, s1, s2, h, i, j, r, v, t, x3, y3, z3;
    f_sqr(&FQ, &p->z, &z1z1);
    f_sqr(&FQ, &q->z, &z2z2);
    f_mul(&FQ, &p->x, &z2z2, &u1);
    f_mul(&FQ, &q->x, &z1z1, &u2);
    f_mul(&FQ, &p->y, &q->z, &s1); f_mul(&FQ, &s1, &z2z2, &s1);
    f_mul(&FQ, &q->y, &p->z, &s2); f_mul(&FQ, &s2, &z1z1, &s2);
    if (fe_eq(&u1, &u2)) {
        if (fe_eq(&s1, &s2)) { g1j_double(p, o); return; }
        g1j_set_id(o); return;
    }
    f_sub(&FQ, &u2, &u1, &h);
    f_dbl(&FQ, &h, &i); f_sqr(&FQ, &i, &i);
    f_mul(&FQ, &h, &i, &j);
    f_sub(&FQ, &s2, &s1, &r); f_dbl(&FQ, &r, &r);
    f_mul(&FQ, &u1, &i, &v);
    f_sqr(&FQ, &r, &x3); f_sub(&FQ, &x3, &j, &x3); f_sub(&FQ, &x3, &v, &x3); f_sub(&FQ, &x3, &v, &x3);
    f_mul(&FQ, &s1, &j, &s1); f_dbl(&FQ, &s1, &s1);
    f_sub(&FQ, &v, &x3, &t); f_mul(&FQ, &r, &t, &y3); f_sub(&FQ, &y3, &s1, &y3);
    f_add(&FQ, &p->z, &q->z, &z3); f_sqr(&FQ, &z3, &z3); f_sub(&FQ, &z3, &z1z1, &z3); f_sub(&FQ, &z3, &z2z2, &z3);
    f_mul(&FQ, &z3, &h, &z3);
    o->x = x3; o->y = y3; o->z = z3;
}
/* madd-2007-bl with the exceptional cases */
static void g1j_add_mixed(const g1j* p, const g1a* q, g1j* o) {
    if (g1a_is_id(q)) { *o = *p; return; }
    if (g1j_is_id(p)) { g1j_from_affine(q, o); return; }
    fe z1z1, u2, s2, h, hh, i, j, r, v, t, x3, y3, z3;
    f_sqr(&FQ, &p->z, &z1z1);
    f_mul(&FQ, &q->x, &z1z1, &u2);
    f_mul(&FQ, &q->y, &z1z1, &s2); f_mul(&FQ, &s2, &p->z, &s2);
    if (fe_eq(&p->x, &u2)) {
        if (fe_eq(&p->y, &s2)) { g1j_double(p, o); return; }
        g1j_set_id(o); return;
    }
    f_sub(&FQ, &u2, &p->x, &h);
    f_sqr(&FQ, &h, &hh);
    f_dbl(&FQ, &hh, &i); f_dbl(&FQ, &i, &i);
    f_mul(&FQ, &h, &i, &j);
    f_sub(&FQ, &s2, &p->y, &r); f_dbl(&FQ, &r, &r);
    f_mul(&FQ, &p->x, &i, &v);
    f_sqr(&FQ, &r, &x3); f_sub(&FQ, &x3, &j, &x3); f_sub(&FQ, &x3, &v, &x3); f_sub(&FQ, &x3, &v, &x3);
    f_mul(&FQ, &p->y, &j, &j); f_dbl(&FQ, &j, &j);
    f_sub(&FQ, &v, &x3, &t); f_mul(&FQ, &r, &t, &y3); f_sub(&FQ, &y3, &j, &y3);
    f_add(&FQ, &p->z, &h, &z3); f_sqr(&FQ, &z3, &z3); f_sub(&FQ, &z3, &z1z1, &z3); f_sub(&FQ, &z3, &hh, &z3);
    o->x = x3; o->y = y3; o->z = z3;
}
static void g1j_to_affine(const g1j* p, g1a* o) {
    if (g1j_is_id(p)) { memset(o, 0, sizeof *o); return; }
    fe zi, zi2, zi3;
    f_inv(&FQ, &p->z, &zi);
    f_sqr(&FQ, &zi, &zi2);
    f_mul(&FQ, &zi2, &zi, &zi3);
    f_mul(&FQ, &p->x, &zi2, &o->x);
    f_mul(&FQ, &p->y, &zi3, &o->y);
}
static void g1j_mul_canon(const g1j* p, const fe* k_canon, g1j* o) {
    g1j acc; g1j_set_id(&acc);
    for (int i = 255; i >= 0; --i) {
        g1j_double(&acc, &acc);
        if ((k_canon->l[i / 64] >> (i % 64)) & 1) g1j_add(&acc, p, &acc);
    }
    *o = acc;
}

void zko_g1_generator(uint64_t out[8]) {
    g1a g; f_from_u64(&FQ, 1, &g.x); f_from_u64(&FQ, 2, &g.y);
    memcpy(out, &g, 64);
}
void zko_g1_double(const uint64_t p[12], uint64_t out[12]) { g1j r; g1j_double((const g1j*)p, &r); memcpy(out, &r, 96); }
void zko_g1_add(const uint64_t p[12], const uint64_t q[12], uint64_t out[12]) {
    g1j r; g1j_add((const g1j*)p, (const g1j*)q, &r); memcpy(out, &r, 96);
}
void zko_g1_add_mixed(const uint64_t p[12], const uint64_t q[8], uint64_t out[12]) {
    g1j r; g1j_add_mixed((const g1j*)p, (const g1a*)q, &r); memcpy(out, &r, 96);
}
void zko_g1_to_affine(const uint64_t p[12], uint64_t out[8]) { g1a r; g1j_to_affine((const g1j*)p, &r); memcpy(out, &r, 64); }
void zko_g1_from_affine(const uint64_t p[8], uint64_t out[12]) { g1j r; g1j_from_affine((const g1a*)p, &r); memcpy(out, &r, 96); }
void zko_g1_mul(const uint64_t p[12], const uint64_t s[4], uint64_t out[12]) {
    fe k; f_from_mont(&FR, FE(s), &k);
    g1j r; g1j_mul_canon((const g1j*)p, &k, &r); memcpy(out, &r, 96);
}
int zko_g1_is_on_curve(const uint64_t p[8]) {
    const g1a* a = (const g1a*)p;
    if (g1a_is_id(a)) return 1;
    fe y2, x3, b;
    f_sqr(&FQ, &a->y, &y2);
    f_sqr(&FQ, &a->x, &x3); f_mul(&FQ, &x3, &a->x, &x3);
    f_from_u64(&FQ, 3, &b); f_add(&FQ, &x3, &b, &x3);
    return fe_eq(&y2, &x3);
}
void zko_g1_to_bytes(const uint64_t p[8], uint8_t out[32]) {
    const g1a* a = (const g1a*)p;
    memset(out, 0, 32);
    if (g1a_is_id(a)) { out[31] |= 0x80; return; }
    fe x, y;
    f_from_mont(&FQ, &a->x, &x);
    f_from_mont(&FQ, &a->y, &y);
    memcpy(out, x.l, 32); /* little-endian host */
    out[31] |= (uint8_t)((y.l[0] & 1) << 6);
}

/* ------------------------------------------------------------------ a3: MSM
 * halo2curves src/msm.rs: multiexp_serial + best_multiexp (unsigned c-bit windows,
 * c = 3 for n < 32 else ceil(ln n); per-window {None, Affine, Projective} buckets). */
typedef struct { int tag; g1a a; g1j p; } bucket; /* tag 0 None, 1 Affine, 2 Projective */

static inline size_t get_at(size_t segment, size_t c, const uint8_t bytes[32]) {
    size_t skip_bits = segment * c, skip_bytes = skip_bits / 8;
    if (skip_bytes >= 32) return 0;
    uint8_t v[8] = {0};
    for (size_t i = 0; i < 8 && skip_bytes + i < 32; ++i) v[i] = bytes[skip_bytes + i];
    uint64_t tmp;
    memcpy(&tmp, v, 8);
    tmp >>= skip_bits - skip_bytes * 8;
    tmp %= (uint64_t)1 << c;
    return (size_t)tmp;
}
static size_t msm_window(size_t n) {
    if (n < 4) return 1;
    if (n < 32) return 3;
    /* (f64::from(n as u32)).ln().ceil() */
    double x = (double)(uint32_t)n, l = 0;
    { extern double log(double); extern double ceil(double); l = ceil(log(x)); }
    return (size_t)l;
}
void zko_multiexp_serial(const uint64_t* coeffs_m, const uint64_t* bases_xy, size_t n, uint64_t acc_io[12]) {
    g1j acc; memcpy(&acc, acc_io, 96);
    const g1a* bases = (const g1a*)bases_xy;
    fe* coeffs = (fe*)malloc(n * sizeof(fe) + 8);
    for (size_t i = 0; i < n; ++i) f_from_mont(&FR, FE(coeffs_m + 4 * i), &coeffs[i]);
    size_t c = msm_window(n);
    size_t segments = 256 / c + 1;
    size_t nb = ((size_t)1 << c) - 1;
    bucket* buckets = (bucket*)malloc(nb * sizeof(bucket));
    for (size_t seg = segments; seg-- > 0;) {
        for (size_t i = 0; i < c; ++i) g1j_double(&acc, &acc);
        for (size_t b = 0; b < nb; ++b) buckets[b].tag = 0;
        for (size_t i = 0; i < n; ++i) {
            size_t d = get_at(seg, c, (const uint8_t*)coeffs[i].l);
            if (d != 0) {
                bucket* B = &buckets[d - 1];
                if (B->tag == 0) { B->tag = 1; B->a = bases[i]; }
                else if (B->tag == 1) { g1j t; g1j_from_affine(&B->a, &t); g1j_add_mixed(&t, &bases[i], &B->p); B->tag = 2; }
                else g1j_add_mixed(&B->p, &bases[i], &B->p);
            }
        }
        g1j running; g1j_set_id(&running);
        for (size_t b = nb; b-- > 0;) {
            if (buckets[b].tag == 1) g1j_add_mixed(&running, &buckets[b].a, &running);
            else if (buckets[b].tag == 2) g1j_add(&running, &buckets[b].p, &running);
            g1j_add(&acc, &running, &acc);
        }
    }
    free(buckets); free(coeffs);
    memcpy(acc_io, &acc, 96);
}
void zko_best_multiexp(const uint64_t* coeffs, const uint64_t* bases_xy, size_t n, int threads, uint64_t out[12]) {
    g1j acc; g1j_set_id(&acc);
    if (threads < 1) threads = 1;
    if (n > (size_t)threads) {
        size_t chunk = n / (size_t)threads;
        size_t nchunks = (n + chunk - 1) / chunk;
        g1j* results = (g1j*)malloc(nchunks * sizeof(g1j));
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
        for (long ci = 0; ci < (long)nchunks; ++ci) {
            size_t lo = (size_t)ci * chunk, hi = lo + chunk > n ? n : lo + chunk;
            g1j_set_id(&results[ci]);
            zko_multiexp_serial(coeffs + 4 * lo, bases_xy + 8 * lo, hi - lo, (uint64_t*)&results[ci]);
        }
        for (size_t ci = 0; ci < nchunks; ++ci) g1j_add(&acc, &results[ci], &acc);
        free(results);
    } else {
        zko_multiexp_serial(coeffs, bases_xy, n, (uint64_t*)&acc);
    }
    memcpy(out, &acc, 96);
}

/* ------------------------------------------------------------------ a5: FFT
 * halo2curves src/fft.rs best_fft: bit-reverse, serial twiddle table, then either the
 * chunked loop (log_n <= log_threads) or recursive_butterfly_arithmetic. */
static size_t bitreverse(size_t n, size_t l) {
    size_t r = 0;
    for (size_t i = 0; i < l; ++i) { r = (r << 1) | (n & 1); n >>= 1; }
    return r;
}
static inline void butterfly0(fe* a, fe* b) {
    fe t = *b;
    *b = *a;
    f_add(&FR, a, &t, a);
    f_sub(&FR, b, &t, b);
}
static inline void butterfly(fe* a, fe* b, const fe* tw) {
    fe t;
    f_mul(&FR, b, tw, &t);
    *b = *a;
    f_add(&FR, a, &t, a);
    f_sub(&FR, b, &t, b);
}
static void recursive_butterfly(fe* a, size_t n, size_t twiddle_chunk, const fe* twiddles, int depth_par) {
    if (n == 2) { butterfly0(&a[0], &a[1]); return; }
    fe* left = a; fe* right = a + n / 2;
    if (depth_par > 0 && n >= 2048) {
#pragma omp task
        recursive_butterfly(left, n / 2, twiddle_chunk * 2, twiddles, depth_par - 1);
#pragma omp task
        recursive_butterfly(right, n / 2, twiddle_chunk * 2, twiddles, depth_par - 1);
#pragma omp taskwait
    } else {
        recursive_butterfly(left, n / 2, twiddle_chunk * 2, twiddles, 0);
        recursive_butterfly(right, n / 2, twiddle_chunk * 2, twiddles, 0);
    }
    butterfly0(&left[0], &right[0]);
    for (size_t i = 1; i < n / 2; ++i) butterfly(&left[i], &right[i], &twiddles[i * twiddle_chunk]);
}
static int log2_floor(int x) { int l = 0; while ((1 << (l + 1)) <= x) ++l; return l; }

static void best_fft_fe(fe* a, const fe* omega, uint32_t log_n, int threads) {
    if (threads < 1) threads = 1;
    size_t n = (size_t)1 << log_n;
    int log_threads = log2_floor(threads);
    for (size_t k = 0; k < n; ++k) {
        size_t rk = bitreverse(k, log_n);
        if (k < rk) { fe t = a[rk]; a[rk] = a[k]; a[k] = t; }
    }
    if (n < 2) return;
    fe* tw = (fe*)malloc((n / 2) * sizeof(fe));
    fe w = FR.one;
    for (size_t i = 0; i < n / 2; ++i) { tw[i] = w; f_mul(&FR, &w, omega, &w); }
    if ((int)log_n <= log_threads) {
        size_t chunk = 2, twiddle_chunk = n / 2;
        for (uint32_t s = 0; s < log_n; ++s) {
            for (size_t base = 0; base < n; base += chunk) {
                fe* left = a + base; fe* right = left + chunk / 2;
                butterfly0(&left[0], &right[0]);
                for (size_t i = 1; i < chunk / 2; ++i) butterfly(&left[i], &right[i], &tw[i * twiddle_chunk]);
            }
            chunk *= 2; twiddle_chunk /= 2;
        }
    } else {
#pragma omp parallel num_threads(threads)
#pragma omp single
        recursive_butterfly(a, n, 1, tw, log_threads + 1);
    }
    free(tw);
}
void zko_best_fft(uint64_t* a, const uint64_t omega[4], uint32_t log_n, int threads) {
    best_fft_fe((fe*)a, FE(omega), log_n, threads);
}

/* ------------------------------------------------------------------ a6: EvaluationDomain
 * halo2_proofs src/poly/domain.rs */
struct zko_domain {
    uint32_t k, extended_k, quotient_poly_degree;
    size_t n, extended_n;
    fe omega, omega_inv, extended_omega, extended_omega_inv, g_coset, g_coset_inv, ifft_divisor, extended_ifft_divisor;
    fe* t_evaluations; size_t n_t;
};
zko_domain* zko_domain_new(uint32_t j, uint32_t k, const uint64_t g_coset[4]) {
    zko_domain* d = (zko_domain*)calloc(1, sizeof *d);
    d->k = k; d->n = (size_t)1 << k;
    d->quotient_poly_degree = j - 1;
    uint32_t ek = k;
    while (((size_t)1 << ek) < d->n * d->quotient_poly_degree) ++ek;
    d->extended_k = ek; d->extended_n = (size_t)1 << ek;
    zko_fr_root_of_unity(ek, d->extended_omega.l);
    d->omega = d->extended_omega;
    for (uint32_t i = k; i < ek; ++i) f_sqr(&FR, &d->omega, &d->omega);
    f_inv(&FR, &d->omega, &d->omega_inv);
    f_inv(&FR, &d->extended_omega, &d->extended_omega_inv);
    if (g_coset) memcpy(d->g_coset.l, g_coset, 32); else f_to_mont(&FR, &FR_ZETA_CANON, &d->g_coset);
    f_sqr(&FR, &d->g_coset, &d->g_coset_inv);
    fe t; f_from_u64(&FR, (uint64_t)1 << k, &t); f_inv(&FR, &t, &d->ifft_divisor);
    f_from_u64(&FR, (uint64_t)1 << ek, &t); f_inv(&FR, &t, &d->extended_ifft_divisor);
    d->n_t = (size_t)1 << (ek - k);
    d->t_evaluations = (fe*)malloc(d->n_t * sizeof(fe));
    uint64_t e[4] = {d->n, 0, 0, 0};
    fe cur, step;
    f_pow(&FR, &d->g_coset, e, &cur);
    f_pow(&FR, &d->extended_omega, e, &step);
    for (size_t i = 0; i < d->n_t; ++i) {
        fe v; f_sub(&FR, &cur, &FR.one, &v);
        f_inv(&FR, &v, &d->t_evaluations[i]);
        f_mul(&FR, &cur, &step, &cur);
    }
    return d;
}
void zko_domain_free(zko_domain* d) { if (d) { free(d->t_evaluations); free(d); } }
uint32_t zko_domain_extended_k(const zko_domain* d) { return d->extended_k; }
uint32_t zko_domain_quotient_poly_degree(const zko_domain* d) { return d->quotient_poly_degree; }
void zko_domain_get(const zko_domain* d, uint64_t omega[4], uint64_t eomega[4], uint64_t g[4]) {
    memcpy(omega, d->omega.l, 32); memcpy(eomega, d->extended_omega.l, 32); memcpy(g, d->g_coset.l, 32);
}
static void ifft_fe(fe* a, const fe* omega_inv, uint32_t log_n, const fe* divisor, int threads) {
    best_fft_fe(a, omega_inv, log_n, threads);
    size_t n = (size_t)1 << log_n;
#pragma omp parallel for num_threads(threads > 0 ? threads : 1)
    for (long i = 0; i < (long)n; ++i) f_mul(&FR, &a[i], divisor, &a[i]);
}
void zko_lagrange_to_coeff(const zko_domain* d, uint64_t* a, int threads) {
    ifft_fe((fe*)a, &d->omega_inv, d->k, &d->ifft_divisor, threads);
}
void zko_coeff_to_lagrange(const zko_domain* d, uint64_t* a, int threads) {
    best_fft_fe((fe*)a, &d->omega, d->k, threads);
}
static void distribute_powers_zeta(const zko_domain* d, fe* a, size_t n, int into_coset, int threads) {
    const fe* cp[2];
    if (into_coset) { cp[0] = &d->g_coset; cp[1] = &d->g_coset_inv; } else { cp[0] = &d->g_coset_inv; cp[1] = &d->g_coset; }
#pragma omp parallel for num_threads(threads > 0 ? threads : 1)
    for (long i = 0; i < (long)n; ++i) {
        size_t j = (size_t)i % 3;
        if (j != 0) f_mul(&FR, &a[i], cp[j - 1], &a[i]);
    }
}
void zko_coeff_to_extended(const zko_domain* d, const uint64_t* coeffs, size_t n_in, uint64_t* out, int threads) {
    fe* a = (fe*)out;
    memcpy(a, coeffs, n_in * 32);
    memset(a + n_in, 0, (d->extended_n - n_in) * 32);
    distribute_powers_zeta(d, a, d->extended_n, 1, threads);
    best_fft_fe(a, &d->extended_omega, d->extended_k, threads);
}
void zko_extended_to_coeff(const zko_domain* d, uint64_t* a, int threads) {
    ifft_fe((fe*)a, &d->extended_omega_inv, d->extended_k, &d->extended_ifft_divisor, threads);
    distribute_powers_zeta(d, (fe*)a, d->extended_n, 0, threads);
}
void zko_divide_by_vanishing_poly(const zko_domain* d, uint64_t* a) {
    fe* v = (fe*)a;
    for (size_t i = 0; i < d->extended_n; ++i) f_mul(&FR, &v[i], &d->t_evaluations[i % d->n_t], &v[i]);
}
/* halo2_proofs src/plonk/keygen.rs: l0, l_blind, l_last -> cosets; l_active_row = 1 - (l_last + l_blind) */
void zko_domain_l_cosets(const zko_domain* d, uint32_t bf, uint64_t* l0_o, uint64_t* l_last_o, uint64_t* l_active_o, int threads) {
    size_t n = d->n, en = d->extended_n;
    fe* tmp = (fe*)calloc(n, sizeof(fe));
    fe* l_blind = (fe*)malloc(en * sizeof(fe));
    tmp[0] = FR.one;
    zko_lagrange_to_coeff(d, (uint64_t*)tmp, threads);
    zko_coeff_to_extended(d, (uint64_t*)tmp, n, l0_o, threads);
    memset(tmp, 0, n * sizeof(fe));
    for (size_t i = 0; i < bf; ++i) tmp[n - 1 - i] = FR.one;
    zko_lagrange_to_coeff(d, (uint64_t*)tmp, threads);
    zko_coeff_to_extended(d, (uint64_t*)tmp, n, (uint64_t*)l_blind, threads);
    memset(tmp, 0, n * sizeof(fe));
    tmp[n - bf - 1] = FR.one;
    zko_lagrange_to_coeff(d, (uint64_t*)tmp, threads);
    zko_coeff_to_extended(d, (uint64_t*)tmp, n, l_last_o, threads);
    fe* ll = (fe*)l_last_o; fe* la = (fe*)l_active_o;
    for (size_t i = 0; i < en; ++i) {
        fe s; f_add(&FR, &ll[i], &l_blind[i], &s);
        f_sub(&FR, &FR.one, &s, &la[i]);
    }
    free(tmp); free(l_blind);
}

/* ------------------------------------------------------------------ a7: evaluate_h
 * halo2_proofs src/plonk/evaluation.rs: GraphEvaluator::evaluate + Evaluator::evaluate_h. */
typedef struct {
    const zk_evalh_args* A;
    const fe *beta, *gamma, *theta, *y;
} eval_ctx;

static inline size_t get_rotation_idx(size_t idx, int32_t rot, int32_t rot_scale, size_t isize) {
    long long v = (long long)idx + (long long)rot * rot_scale;
    long long m = (long long)isize;
    v %= m; if (v < 0) v += m;
    return (size_t)v;
}
static inline const fe* vs_get(const eval_ctx* C, const zk_graph* g, const int32_t* vs, const size_t* rots,
                               const fe* inter, const fe* prev) {
    const zk_evalh_args* A = C->A;
    switch (vs[0]) {
        case ZK_VS_CONSTANT: return FE(g->constants + 4 * (size_t)vs[1]);
        case ZK_VS_INTERMEDIATE: return &inter[vs[1]];
        case ZK_VS_FIXED: return FE(A->fixed_cosets[vs[1]] + 4 * rots[vs[2]]);
        case ZK_VS_ADVICE: return FE(A->advice_cosets[vs[1]] + 4 * rots[vs[2]]);
        case ZK_VS_INSTANCE: return FE(A->instance_cosets[vs[1]] + 4 * rots[vs[2]]);
        case ZK_VS_CHALLENGE: return FE(A->challenges + 4 * (size_t)vs[1]);
        case ZK_VS_BETA: return C->beta;
        case ZK_VS_GAMMA: return C->gamma;
        case ZK_VS_THETA: return C->theta;
        case ZK_VS_Y: return C->y;
        case ZK_VS_PREVIOUS: return prev;
    }
    return NULL;
}
static int graph_evaluate(const eval_ctx* C, const zk_graph* g, size_t idx, int32_t rot_scale, size_t isize,
                          const fe* prev, size_t* rots, fe* inter, fe* out) {
    for (uint32_t r = 0; r < g->n_rotations; ++r) rots[r] = get_rotation_idx(idx, g->rotations[r], rot_scale, isize);
    const int32_t* pc = g->code;
    int32_t last_target = -1;
    for (uint32_t ci = 0; ci < g->n_calculations; ++ci) {
        int32_t op = pc[0], target = pc[1], nsrc = pc[2];
        const int32_t* src = pc + 3;
        fe res;
#define GET(i) vs_get(C, g, src + 3 * (i), rots, inter, prev)
        switch (op) {
            case ZK_OP_ADD: f_add(&FR, GET(0), GET(1), &res); break;
            case ZK_OP_SUB: f_sub(&FR, GET(0), GET(1), &res); break;
            case ZK_OP_MUL: f_mul(&FR, GET(0), GET(1), &res); break;
            case ZK_OP_SQUARE: f_sqr(&FR, GET(0), &res); break;
            case ZK_OP_DOUBLE: f_dbl(&FR, GET(0), &res); break;
            case ZK_OP_NEGATE: f_neg(&FR, GET(0), &res); break;
            case ZK_OP_STORE: res = *GET(0); break;
            case ZK_OP_HORNER: {
                res = *GET(0);
                const fe* factor = GET(1);
                for (int32_t p = 2; p < nsrc; ++p) { f_mul(&FR, &res, factor, &res); f_add(&FR, &res, GET(p), &res); }
                break;
            }
            default: return -1;
        }
#undef GET
        inter[target] = res;
        last_target = target;
        pc += 3 + 3 * nsrc;
    }
    if (last_target >= 0) *out = inter[last_target]; else memset(out, 0, sizeof *out);
    return 0;
}
static uint32_t max_u32(uint32_t a, uint32_t b) { return a > b ? a : b; }

int zko_evaluate_h(const zk_evalh_args* A, uint64_t* out_u, int threads) {
    if (threads < 1) threads = 1;
    fe* values = (fe*)out_u;
    size_t isize = (size_t)1 << A->extended_k;
    int32_t rot_scale = 1 << (A->extended_k - A->k);
    eval_ctx C = {A, FE(A->beta), FE(A->gamma), FE(A->theta), FE(A->y)};
    const fe* l0 = FE(A->l0); const fe* l_last = FE(A->l_last); const fe* l_active = FE(A->l_active_row);
    const fe* y = C.y; const fe* beta = C.beta; const fe* gamma = C.gamma;
    int err = 0;
    uint32_t max_rot = A->custom_gates.n_rotations, max_int = A->custom_gates.n_intermediates;
    for (uint32_t i = 0; i < A->n_lookups; ++i) {
        max_rot = max_u32(max_rot, A->lookup_graphs[i].n_rotations);
        max_int = max_u32(max_int, A->lookup_graphs[i].n_intermediates);
    }
    memset(values, 0, isize * sizeof(fe));

    /* custom gates */
#pragma omp parallel num_threads(threads)
    {
        size_t* rots = (size_t*)malloc((max_rot + 1) * sizeof(size_t));
        fe* inter = (fe*)malloc((max_int + 1) * sizeof(fe));
#pragma omp for
        for (long idx = 0; idx < (long)isize; ++idx) {
            fe prev = values[idx];
            if (graph_evaluate(&C, &A->custom_gates, (size_t)idx, rot_scale, isize, &prev, rots, inter, &values[idx])) err = -1;
        }
        free(rots); free(inter);
    }
    if (err) return err;

    /* permutation argument */
    if (A->n_perm_sets > 0) {
        uint32_t bf = A->blinding_factors;
        int32_t last_rotation = -((int32_t)bf + 1);
        uint32_t chunk_len = A->cs_degree - 2;
        fe delta_start; f_mul(&FR, beta, FE(A->g_coset), &delta_start);
        const fe* first_set = FE(A->perm_product_cosets[0]);
        const fe* last_set = FE(A->perm_product_cosets[A->n_perm_sets - 1]);
#pragma omp parallel num_threads(threads)
        {
#ifdef _OPENMP
        size_t nth = (size_t)omp_get_num_threads(), tid = (size_t)omp_get_thread_num();
#else
        size_t nth = 1, tid = 0;
#endif
        /* parallelize(): contiguous chunks; beta_term = extended_omega^start, then *= extended_omega per row */
        size_t chunk = (isize + nth - 1) / nth, start = tid * chunk, end = start + chunk > isize ? isize : start + chunk;
        uint64_t e[4] = {start, 0, 0, 0};
        fe beta_term; f_pow(&FR, FE(A->extended_omega), e, &beta_term);
        for (size_t idx = start; idx < end; ++idx) {
            fe v = values[idx], t, u;
            size_t r_next = get_rotation_idx(idx, 1, rot_scale, isize);
            size_t r_last = get_rotation_idx(idx, last_rotation, rot_scale, isize);
            /* l_0(X) * (1 - z_0(X)) */
            f_mul(&FR, &v, y, &v); f_sub(&FR, &FR.one, &first_set[idx], &t); f_mul(&FR, &t, &l0[idx], &t); f_add(&FR, &v, &t, &v);
            /* l_last(X) * (z_l(X)^2 - z_l(X)) */
            f_mul(&FR, &v, y, &v); f_sqr(&FR, &last_set[idx], &t); f_sub(&FR, &t, &last_set[idx], &t);
            f_mul(&FR, &t, &l_last[idx], &t); f_add(&FR, &v, &t, &v);
            /* l_0(X) * (z_i(X) - z_{i-1}(\omega^(last) X)) */
            for (uint32_t s = 1; s < A->n_perm_sets; ++s) {
                f_mul(&FR, &v, y, &v);
                f_sub(&FR, FE(A->perm_product_cosets[s]) + idx, FE(A->perm_product_cosets[s - 1]) + r_last, &t);
                f_mul(&FR, &t, &l0[idx], &t); f_add(&FR, &v, &t, &v);
            }
            fe current_delta; f_mul(&FR, &delta_start, &beta_term, &current_delta);
            for (uint32_t s = 0; s < A->n_perm_sets; ++s) {
                const fe* set = FE(A->perm_product_cosets[s]);
                uint32_t c0 = s * chunk_len, c1 = c0 + chunk_len; if (c1 > A->n_perm_columns) c1 = A->n_perm_columns;
                fe left = set[r_next], right = set[idx];
                for (uint32_t c = c0; c < c1; ++c) {
                    const uint64_t* col = A->perm_column_type[c] == 0 ? A->advice_cosets[A->perm_column_index[c]]
                                        : A->perm_column_type[c] == 1 ? A->fixed_cosets[A->perm_column_index[c]]
                                                                      : A->instance_cosets[A->perm_column_index[c]];
                    const fe* val = FE(col) + idx;
                    f_mul(&FR, beta, FE(A->perm_sigma_cosets[c]) + idx, &t); f_add(&FR, val, &t, &t); f_add(&FR, &t, gamma, &t);
                    f_mul(&FR, &left, &t, &left);
                }
                for (uint32_t c = c0; c < c1; ++c) {
                    const uint64_t* col = A->perm_column_type[c] == 0 ? A->advice_cosets[A->perm_column_index[c]]
                                        : A->perm_column_type[c] == 1 ? A->fixed_cosets[A->perm_column_index[c]]
                                                                      : A->instance_cosets[A->perm_column_index[c]];
                    const fe* val = FE(col) + idx;
                    f_add(&FR, val, &current_delta, &u); f_add(&FR, &u, gamma, &u);
                    f_mul(&FR, &right, &u, &right);
                    f_mul(&FR, &current_delta, FE(A->delta), &current_delta);
                }
                f_mul(&FR, &v, y, &v); f_sub(&FR, &left, &right, &t); f_mul(&FR, &t, &l_active[idx], &t); f_add(&FR, &v, &t, &v);
            }
            values[idx] = v;
            f_mul(&FR, &beta_term, FE(A->extended_omega), &beta_term);
        }
        }
    }

    /* lookups */
    for (uint32_t n = 0; n < A->n_lookups; ++n) {
        const zk_graph* g = &A->lookup_graphs[n];
        const fe* prod = FE(A->lookup_product_cosets[n]);
        const fe* pin = FE(A->lookup_input_cosets[n]);
        const fe* ptab = FE(A->lookup_table_cosets[n]);
#pragma omp parallel num_threads(threads)
        {
            size_t* rots = (size_t*)malloc((max_rot + 1) * sizeof(size_t));
            fe* inter = (fe*)malloc((max_int + 1) * sizeof(fe));
            fe zero; memset(&zero, 0, sizeof zero);
#pragma omp for
            for (long idx_l = 0; idx_l < (long)isize; ++idx_l) {
                size_t idx = (size_t)idx_l;
                fe table_value;
                if (graph_evaluate(&C, g, idx, rot_scale, isize, &zero, rots, inter, &table_value)) { err = -1; continue; }
                size_t r_next = get_rotation_idx(idx, 1, rot_scale, isize);
                size_t r_prev = get_rotation_idx(idx, -1, rot_scale, isize);
                fe a_minus_s, v = values[idx], t, u;
                f_sub(&FR, &pin[idx], &ptab[idx], &a_minus_s);
                /* l_0(X) * (1 - z(X)) */
                f_mul(&FR, &v, y, &v); f_sub(&FR, &FR.one, &prod[idx], &t); f_mul(&FR, &t, &l0[idx], &t); f_add(&FR, &v, &t, &v);
                /* l_last(X) * (z(X)^2 - z(X)) */
                f_mul(&FR, &v, y, &v); f_sqr(&FR, &prod[idx], &t); f_sub(&FR, &t, &prod[idx], &t);
                f_mul(&FR, &t, &l_last[idx], &t); f_add(&FR, &v, &t, &v);
                /* (z(wX)(a'(X)+beta)(s'(X)+gamma) - z(X)*table_value) * l_active */
                f_add(&FR, &pin[idx], beta, &t); f_add(&FR, &ptab[idx], gamma, &u); f_mul(&FR, &t, &u, &t);
                f_mul(&FR, &prod[r_next], &t, &t);
                f_mul(&FR, &prod[idx], &table_value, &u); f_sub(&FR, &t, &u, &t);
                f_mul(&FR, &t, &l_active[idx], &t);
                f_mul(&FR, &v, y, &v); f_add(&FR, &v, &t, &v);
                /* l_0(X) * (a'(X) - s'(X)) */
                f_mul(&FR, &v, y, &v); f_mul(&FR, &a_minus_s, &l0[idx], &t); f_add(&FR, &v, &t, &v);
                /* (a'(X)-s'(X)) * (a'(X)-a'(w^-1 X)) * l_active */
                f_sub(&FR, &pin[idx], &pin[r_prev], &t); f_mul(&FR, &a_minus_s, &t, &t); f_mul(&FR, &t, &l_active[idx], &t);
                f_mul(&FR, &v, y, &v); f_add(&FR, &v, &t, &v);
                values[idx] = v;
            }
            free(rots); free(inter);
        }
        if (err) return err;
    }
    return 0;
}

void zko_eval_polynomial(const uint64_t* coeffs, size_t n, const uint64_t x[4], uint64_t out[4]);
/* ------------------------------------------------------------------ a8: grand products
 * halo2_proofs src/plonk/permutation/prover.rs (Argument::commit), src/plonk/lookup/prover.rs (commit_product),
 * ff::BatchInvert.  The random blinding rows are an input here (upstream draws them from its rng). */
void zko_batch_invert(uint64_t* a_u, size_t n) {
    fe* a = (fe*)a_u;
    fe* pre = (fe*)malloc((n + 1) * sizeof(fe));
    fe acc = FR.one;
    for (size_t i = 0; i < n; ++i) {
        pre[i] = acc;
        if (!fe_is_zero(&a[i])) f_mul(&FR, &acc, &a[i], &acc);
    }
    fe inv; f_inv(&FR, &acc, &inv);
    for (size_t i = n; i-- > 0;) {
        if (fe_is_zero(&a[i])) continue;
        fe t; f_mul(&FR, &inv, &pre[i], &t);
        f_mul(&FR, &inv, &a[i], &inv);
        a[i] = t;
    }
    free(pre);
}
void zko_permutation_products(uint32_t k, uint32_t n_cols, uint32_t chunk_len, const uint64_t* const* values, const uint64_t* const* sigmas,
                              const uint64_t beta_u[4], const uint64_t gamma_u[4], uint32_t bf, const uint64_t* blinding_rand,
                              uint64_t* const* z_out, int threads) {
    (void)threads;
    size_t n = (size_t)1 << k;
    const fe* beta = FE(beta_u); const fe* gamma = FE(gamma_u);
    fe omega; zko_fr_root_of_unity(k, omega.l);
    fe delta; f_to_mont(&FR, &FR_DELTA_CANON, &delta);
    fe deltaomega = FR.one, last_z = FR.one;
    fe* mv = (fe*)malloc(n * sizeof(fe));
    uint32_t nsets = (n_cols + chunk_len - 1) / chunk_len;
    for (uint32_t s = 0; s < nsets; ++s) {
        uint32_t c0 = s * chunk_len, c1 = c0 + chunk_len > n_cols ? n_cols : c0 + chunk_len;
        for (size_t i = 0; i < n; ++i) mv[i] = FR.one;
        for (uint32_t c = c0; c < c1; ++c)
            for (size_t i = 0; i < n; ++i) {
                fe t; f_mul(&FR, beta, FE(sigmas[c] + 4 * i), &t); f_add(&FR, &t, gamma, &t); f_add(&FR, &t, FE(values[c] + 4 * i), &t);
                f_mul(&FR, &mv[i], &t, &mv[i]);
            }
        zko_batch_invert((uint64_t*)mv, n);
        for (uint32_t c = c0; c < c1; ++c) {
            fe dw = deltaomega;
            for (size_t i = 0; i < n; ++i) {
                fe t; f_mul(&FR, &dw, beta, &t); f_add(&FR, &t, gamma, &t); f_add(&FR, &t, FE(values[c] + 4 * i), &t);
                f_mul(&FR, &mv[i], &t, &mv[i]);
                f_mul(&FR, &dw, &omega, &dw);
            }
            f_mul(&FR, &deltaomega, &delta, &deltaomega);
        }
        fe* z = (fe*)z_out[s];
        z[0] = last_z;
        for (size_t row = 1; row < n; ++row) f_mul(&FR, &z[row - 1], &mv[row - 1], &z[row]);
        for (uint32_t j = 0; j < bf; ++j) memcpy(&z[n - bf + j], blinding_rand + 4 * ((size_t)s * bf + j), 32);
        last_z = z[n - (bf + 1)];
    }
    free(mv);
}
void zko_lookup_product(uint32_t k, const uint64_t* compressed_input, const uint64_t* compressed_table, const uint64_t* permuted_input,
                        const uint64_t* permuted_table, const uint64_t beta_u[4], const uint64_t gamma_u[4], uint32_t bf,
                        const uint64_t* blinding_rand, uint64_t* z_out) {
    size_t n = (size_t)1 << k;
    const fe* beta = FE(beta_u); const fe* gamma = FE(gamma_u);
    fe* lp = (fe*)malloc(n * sizeof(fe));
    for (size_t i = 0; i < n; ++i) {
        fe a, b; f_add(&FR, beta, FE(permuted_input + 4 * i), &a); f_add(&FR, gamma, FE(permuted_table + 4 * i), &b);
        f_mul(&FR, &a, &b, &lp[i]);
    }
    zko_batch_invert((uint64_t*)lp, n);
    for (size_t i = 0; i < n; ++i) {
        fe a, b; f_add(&FR, FE(compressed_input + 4 * i), beta, &a); f_add(&FR, FE(compressed_table + 4 * i), gamma, &b);
        f_mul(&FR, &lp[i], &a, &lp[i]); f_mul(&FR, &lp[i], &b, &lp[i]);
    }
    fe* z = (fe*)z_out;
    fe state = FR.one;
    z[0] = FR.one;   /* once(ONE).chain(lookup_product).scan(ONE, *=): first output is ONE * ONE */
    for (size_t i = 1; i < n - bf; ++i) { f_mul(&FR, &state, &lp[i - 1], &state); z[i] = state; }
    for (uint32_t j = 0; j < bf; ++j) memcpy(&z[n - bf + j], blinding_rand + 4 * j, 32);
    free(lp);
}
/* lookup::prover::permute_expression_pair (halo2_proofs src/plonk/lookup/prover.rs): sort the input column over the
 * usable rows, put each distinct input value in the table column at its first row, fill the repeated rows with the
 * leftover table values (ascending value -> descending row), append the blinding rows.  -1 = ConstraintSystemFailure. */
typedef struct { fe canon, mont; } pe_item;
static int pe_cmp(const void* a, const void* b) {
    const pe_item* x = (const pe_item*)a; const pe_item* y = (const pe_item*)b;
    for (int i = 3; i >= 0; --i) {
        if (x->canon.l[i] < y->canon.l[i]) return -1;
        if (x->canon.l[i] > y->canon.l[i]) return 1;
    }
    return 0;
}
int zko_permute_expression_pair(uint32_t k, uint32_t bf, const uint64_t* input, const uint64_t* table, const uint64_t* blind_in,
                                const uint64_t* blind_tab, uint64_t* perm_in, uint64_t* perm_tab) {
    size_t n = (size_t)1 << k, u = n - (bf + 1);
    pe_item* A = (pe_item*)malloc(u * sizeof(pe_item));
    pe_item* T = (pe_item*)malloc(u * sizeof(pe_item));
    for (size_t i = 0; i < u; ++i) {
        A[i].mont = *FE(input + 4 * i); f_from_mont(&FR, &A[i].mont, &A[i].canon);
        T[i].mont = *FE(table + 4 * i); f_from_mont(&FR, &T[i].mont, &T[i].canon);
    }
    qsort(A, u, sizeof(pe_item), pe_cmp);
    qsort(T, u, sizeof(pe_item), pe_cmp);   /* BTreeMap iteration order, with multiplicity */
    char* removed = (char*)calloc(u, 1);
    size_t* repeated = (size_t*)malloc(u * sizeof(size_t));
    size_t nrep = 0;
    int rc = 0;
    fe* pin = (fe*)perm_in; fe* ptab = (fe*)perm_tab;
    for (size_t row = 0; row < u && rc == 0; ++row) {
        pin[row] = A[row].mont;
        if (row == 0 || pe_cmp(&A[row], &A[row - 1]) != 0) {
            ptab[row] = A[row].mont;
            size_t lo = 0, hi = u;   /* first table entry >= value */
            while (lo < hi) { size_t mid = (lo + hi) / 2; if (pe_cmp(&T[mid], &A[row]) < 0) lo = mid + 1; else hi = mid; }
            if (lo >= u || pe_cmp(&T[lo], &A[row]) != 0) rc = -1; else removed[lo] = 1;
        } else {
            repeated[nrep++] = row;
        }
    }
    if (rc == 0) {
        for (size_t t = 0; t < u; ++t) {
            if (removed[t]) continue;
            ptab[repeated[--nrep]] = T[t].mont;
        }
        if (nrep != 0) rc = -1;
        for (uint32_t j = 0; j <= bf; ++j) { memcpy(&pin[u + j], blind_in + 4 * j, 32); memcpy(&ptab[u + j], blind_tab + 4 * j, 32); }
    }
    free(A); free(T); free(removed); free(repeated);
    return rc;
}
/* eval_polynomial for a batch (halo2_proofs src/arithmetic.rs) */
void zko_eval_polynomials(const uint64_t* const* polys, size_t npolys, size_t n, const uint64_t x[4], uint64_t* out) {
    for (size_t j = 0; j < npolys; ++j) zko_eval_polynomial(polys[j], n, x, out + 4 * j);
}

/* ------------------------------------------------------------------ SHPLONK's polynomial arithmetic
 * halo2_proofs src/poly/kzg/multiopen/shplonk/prover.rs [UPSTREAM-RECALL]: the prover forms linear combinations of
 * coefficient-form polynomials (powers of y within a rotation set, of v across sets), subtracts the low-degree
 * interpolant, and divides by prod (X - r) with arithmetic::kate_division (synthetic division, remainder dropped). */
void zko_linear_combination(const uint64_t* const* polys, size_t npolys, size_t n, const uint64_t* coeffs, const uint64_t* low, size_t nlow,
                            uint64_t* out) {
    for (size_t i = 0; i < n; ++i) {
        fe acc; memset(&acc, 0, sizeof acc);
        for (size_t j = 0; j < npolys; ++j) {
            fe t; f_mul(&FR, FE(polys[j] + 4 * i), FE(coeffs + 4 * j), &t);
            f_add(&FR, &acc, &t, &acc);
        }
        if (i < nlow) f_sub(&FR, &acc, FE(low + 4 * i), &acc);
        memcpy(out + 4 * i, acc.l, 32);
    }
}
/* a <- a / (X - root), in place: q[j] = s[j+1] with s[j] = a[j] + root * s[j+1]; q[n-1] = 0 (upstream returns n-1 coefficients and the
 * caller resizes to n) */
void zko_kate_division(uint64_t* a, size_t n, const uint64_t root[4]) {
    fe s; memset(&s, 0, sizeof s);
    for (size_t j = n; j-- > 0;) {
        fe aj; memcpy(aj.l, a + 4 * j, 32);
        memcpy(a + 4 * j, s.l, 32);
        f_mul(&FR, &s, FE(root), &s);
        f_add(&FR, &s, &aj, &s);
    }
}

/* ------------------------------------------------------------------ synthetic data */
uint64_t zko_splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
void zko_synth_raw253(uint64_t seed, uint64_t idx, uint64_t out[4]) {
    for (uint64_t j = 0; j < 4; ++j) out[j] = zko_splitmix64(seed + (idx * 4 + j) * 0x2545F4914F6CDD1DULL);
    out[3] &= 0x1FFFFFFFFFFFFFFFULL;
}

void zko_synth_fill(uint64_t seed, uint64_t first, size_t n, uint64_t* out) {
    for (size_t i = 0; i < n; ++i) zko_synth_raw253(seed, first + i, out + 4 * i);
}

/* [k]G by an 8-bit windowed fixed-base table (32 windows x 255 entries), batch-normalised. */
void zko_fixed_base_mul(const uint64_t* scalars, size_t n, uint64_t* out_xy, int threads) {
    if (threads < 1) threads = 1;
    g1a* table = (g1a*)malloc(32 * 256 * sizeof(g1a));
    g1a gen; zko_g1_generator((uint64_t*)&gen);
    g1j base; g1j_from_affine(&gen, &base);
    for (int w = 0; w < 32; ++w) {
        g1j acc; g1j_set_id(&acc);
        memset(&table[w * 256], 0, sizeof(g1a));
        for (int d = 1; d < 256; ++d) {
            g1j_add(&acc, &base, &acc);
            g1j_to_affine(&acc, &table[w * 256 + d]);
        }
        g1j_add(&acc, &base, &base); /* base <- 256 * base */
    }
    g1a* out = (g1a*)out_xy;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 64)
    for (long i = 0; i < (long)n; ++i) {
        fe k; f_from_mont(&FR, FE(scalars + 4 * i), &k);
        const uint8_t* kb = (const uint8_t*)k.l;
        g1j acc; g1j_set_id(&acc);
        for (int w = 0; w < 32; ++w) if (kb[w]) g1j_add_mixed(&acc, &table[w * 256 + kb[w]], &acc);
        g1j_to_affine(&acc, &out[i]);
    }
    free(table);
}
void zko_kzg_setup_scalars(uint32_t k, const uint64_t s_u[4], uint64_t* monomial, uint64_t* lagrange) {
    size_t n = (size_t)1 << k;
    const fe* s = FE(s_u);
    fe* mono = (fe*)monomial; fe* lag = (fe*)lagrange;
    fe cur = FR.one;
    for (size_t i = 0; i < n; ++i) { mono[i] = cur; f_mul(&FR, &cur, s, &cur); }
    /* l_i(s) = (s^n - 1)/n * w^i / (s - w^i) */
    fe omega; zko_fr_root_of_unity(k, omega.l);
    fe sn = cur, num, nfe, ninv;
    f_sub(&FR, &sn, &FR.one, &num);
    f_from_u64(&FR, (uint64_t)n, &nfe); f_inv(&FR, &nfe, &ninv); f_mul(&FR, &num, &ninv, &num);
    fe w = FR.one;
    for (size_t i = 0; i < n; ++i) {
        fe d, di; f_sub(&FR, s, &w, &d); f_inv(&FR, &d, &di);
        f_mul(&FR, &num, &w, &lag[i]); f_mul(&FR, &lag[i], &di, &lag[i]);
        f_mul(&FR, &w, &omega, &w);
    }
}
void zko_eval_polynomial(const uint64_t* coeffs, size_t n, const uint64_t x[4], uint64_t out[4]) {
    fe acc; memset(&acc, 0, sizeof acc);
    for (size_t i = n; i-- > 0;) { f_mul(&FR, &acc, FE(x), &acc); f_add(&FR, &acc, FE(coeffs + 4 * i), &acc); }
    memcpy(out, acc.l, 32);
}
