/* zkoracle.h — CPU restatement of the create_proof hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * parity unpinned: /root/reference holds no golden vector for this path and the arithmetic it
 * calls (gen_snark_shplonk at src/helpers.rs:233,299, src/bin/cli.rs:320,343,369,462;
 * gen_evm_proof_shplonk at src/bin/cli.rs:519) lives in un-vendored crates pinned by
 * Cargo.lock:1320-1322 (halo2_proofs, axiom fork @ 4b42325) and Cargo.lock:1359-1361
 * (halo2curves 0.4.0 @ e185711).  No Rust toolchain exists here, so those crates cannot be
 * built; this file restates their published algorithms ([UPSTREAM-RECALL]) and
 * is pinned against first-principles big-int vectors (oracle/pyref.py -> tests/golden/).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (libzkhip.so) never links or calls it.
 *
 * All field elements cross this API as 4 little-endian u64 limbs in Montgomery form
 * (R = 2^256), the in-memory layout of halo2curves bn256::{Fr,Fq}.  Affine points are
 * {x, y} (8 u64, identity = all zero); Jacobian points are {x, y, z} (12 u64, identity z=0).
 */
#ifndef ZKORACLE_H
#define ZKORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- a1: fields (halo2curves src/bn256/{fr,fq}.rs) ---- */
void zko_fr_mul(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]);
void zko_fr_add(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]);
void zko_fr_sub(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]);
void zko_fr_inv(const uint64_t a[4], uint64_t out[4]);            /* 0 -> 0 */
void zko_fr_pow(const uint64_t a[4], const uint64_t e[4], uint64_t out[4]);
void zko_fr_to_repr(const uint64_t a[4], uint64_t out[4]);        /* Montgomery -> canonical */
void zko_fr_from_repr(const uint64_t a[4], uint64_t out[4]);      /* canonical -> Montgomery */
void zko_fq_mul(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]);
void zko_fq_add(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]);
void zko_fq_sub(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]);
void zko_fq_inv(const uint64_t a[4], uint64_t out[4]);
void zko_fq_to_repr(const uint64_t a[4], uint64_t out[4]);
void zko_fq_from_repr(const uint64_t a[4], uint64_t out[4]);
void zko_fr_root_of_unity(uint32_t k, uint64_t out[4]);           /* omega_k, Montgomery */
void zko_fr_constants(uint64_t zeta[4], uint64_t delta[4]);       /* Fr::ZETA, Fr::DELTA */

/* ---- a2: G1 (halo2curves src/bn256/curve.rs, src/derive/curve.rs) ---- */
void zko_g1_generator(uint64_t out_xy[8]);
void zko_g1_double(const uint64_t p[12], uint64_t out[12]);
void zko_g1_add(const uint64_t p[12], const uint64_t q[12], uint64_t out[12]);
void zko_g1_add_mixed(const uint64_t p[12], const uint64_t q_xy[8], uint64_t out[12]);
void zko_g1_to_affine(const uint64_t p[12], uint64_t out_xy[8]);
void zko_g1_from_affine(const uint64_t p_xy[8], uint64_t out[12]);
void zko_g1_mul(const uint64_t p[12], const uint64_t scalar_mont[4], uint64_t out[12]);
int  zko_g1_is_on_curve(const uint64_t p_xy[8]);
void zko_g1_to_bytes(const uint64_t p_xy[8], uint8_t out[32]);    /* compressed, see pyref.compress */

/* ---- a3/a4: MSM (halo2curves src/msm.rs best_multiexp / multiexp_serial) ----
 * threads = rayon pool size being restated (chunks = n / threads, partials folded in order). */
void zko_multiexp_serial(const uint64_t* coeffs, const uint64_t* bases_xy, size_t n, uint64_t acc[12]);
void zko_best_multiexp(const uint64_t* coeffs, const uint64_t* bases_xy, size_t n, int threads,
                       uint64_t out[12]);

/* ---- a5: FFT (halo2curves src/fft.rs best_fft) ---- */
void zko_best_fft(uint64_t* a, const uint64_t omega[4], uint32_t log_n, int threads);

/* ---- a6: EvaluationDomain (halo2_proofs src/poly/domain.rs) ---- */
typedef struct zko_domain zko_domain;
zko_domain* zko_domain_new(uint32_t j, uint32_t k, const uint64_t g_coset[4] /* NULL = Fr::ZETA */);
void zko_domain_free(zko_domain*);
uint32_t zko_domain_extended_k(const zko_domain*);
uint32_t zko_domain_quotient_poly_degree(const zko_domain*);
void zko_domain_get(const zko_domain*, uint64_t omega[4], uint64_t extended_omega[4], uint64_t g_coset[4]);
void zko_lagrange_to_coeff(const zko_domain*, uint64_t* a /* n */, int threads);
void zko_coeff_to_lagrange(const zko_domain*, uint64_t* a /* n */, int threads);
/* in: n_in coefficients (<= extended n); out: extended_n evaluations */
void zko_coeff_to_extended(const zko_domain*, const uint64_t* coeffs, size_t n_in, uint64_t* out, int threads);
/* in/out: extended_n values in place; the first n*quotient_poly_degree entries are the result */
void zko_extended_to_coeff(const zko_domain*, uint64_t* a, int threads);
void zko_divide_by_vanishing_poly(const zko_domain*, uint64_t* a /* extended_n */);
/* l_0, l_last, l_active_row on the extended coset (keygen: plonk/keygen.rs), each extended_n */
void zko_domain_l_cosets(const zko_domain*, uint32_t blinding_factors, uint64_t* l0, uint64_t* l_last,
                         uint64_t* l_active, int threads);

/* ---- a7: plonk::evaluation (halo2_proofs src/plonk/evaluation.rs) ----
 * ValueSource = 3 x int32 {kind, a, b}; kinds: */
enum { ZK_VS_CONSTANT = 0, ZK_VS_INTERMEDIATE = 1, ZK_VS_FIXED = 2, ZK_VS_ADVICE = 3, ZK_VS_INSTANCE = 4,
       ZK_VS_CHALLENGE = 5, ZK_VS_BETA = 6, ZK_VS_GAMMA = 7, ZK_VS_THETA = 8, ZK_VS_Y = 9,
       ZK_VS_PREVIOUS = 10 };
/* Calculation record in the int32 code stream: {op, target, nsrc, nsrc x ValueSource}.
 * ADD/SUB/MUL: (a, b); SQUARE/DOUBLE/NEGATE/STORE: (a); HORNER: (start, factor, parts...). */
enum { ZK_OP_ADD = 0, ZK_OP_SUB = 1, ZK_OP_MUL = 2, ZK_OP_SQUARE = 3, ZK_OP_DOUBLE = 4, ZK_OP_NEGATE = 5,
       ZK_OP_HORNER = 6, ZK_OP_STORE = 7 };

typedef struct {
    const uint64_t* constants;   /* n_constants x 4, Montgomery */
    const int32_t*  rotations;   /* n_rotations */
    const int32_t*  code;        /* n_code_words int32 */
    uint32_t n_constants, n_rotations, n_code_words, n_calculations, n_intermediates;
} zk_graph;

typedef struct {
    uint32_t k, extended_k, cs_degree, blinding_factors;
    uint64_t extended_omega[4], g_coset[4], delta[4];
    uint64_t beta[4], gamma[4], theta[4], y[4];
    uint32_t n_fixed, n_advice, n_instance, n_challenges;
    const uint64_t* const* fixed_cosets;     /* each extended_n x 4 */
    const uint64_t* const* advice_cosets;
    const uint64_t* const* instance_cosets;
    const uint64_t* challenges;              /* n_challenges x 4 */
    const uint64_t* l0; const uint64_t* l_last; const uint64_t* l_active_row;
    zk_graph custom_gates;
    uint32_t n_perm_columns, n_perm_sets;
    const uint32_t* perm_column_type;        /* 0 advice, 1 fixed, 2 instance */
    const uint32_t* perm_column_index;
    const uint64_t* const* perm_sigma_cosets;    /* n_perm_columns */
    const uint64_t* const* perm_product_cosets;  /* n_perm_sets */
    uint32_t n_lookups, _pad;
    const zk_graph* lookup_graphs;                   /* n_lookups */
    const uint64_t* const* lookup_product_cosets;
    const uint64_t* const* lookup_input_cosets;      /* permuted input A' */
    const uint64_t* const* lookup_table_cosets;      /* permuted table S' */
} zk_evalh_args;

/* out: extended_n x 4.  Returns 0, or -1 on a malformed program. */
int zko_evaluate_h(const zk_evalh_args* args, uint64_t* out, int threads);

/* ---- a8 (next rows): grand products and evaluations ---- */
void zko_batch_invert(uint64_t* a, size_t n);   /* zeros stay zero (ff::BatchInvert) */
/* permutation::prover::commit: z polynomials (Lagrange form, blinding rows = blinding_rand[set][bf]) */
void zko_permutation_products(uint32_t k, uint32_t n_cols, uint32_t chunk_len, const uint64_t* const* values,
                              const uint64_t* const* sigmas, const uint64_t beta[4], const uint64_t gamma[4], uint32_t blinding_factors,
                              const uint64_t* blinding_rand, uint64_t* const* z_out, int threads);
/* lookup::prover::commit_product */
void zko_lookup_product(uint32_t k, const uint64_t* compressed_input, const uint64_t* compressed_table, const uint64_t* permuted_input,
                        const uint64_t* permuted_table, const uint64_t beta[4], const uint64_t gamma[4], uint32_t blinding_factors,
                        const uint64_t* blinding_rand, uint64_t* z_out);
/* lookup::prover::permute_expression_pair; blind_* hold blinding_factors + 1 elements each; -1 = ConstraintSystemFailure */
int zko_permute_expression_pair(uint32_t k, uint32_t blinding_factors, const uint64_t* input, const uint64_t* table,
                                const uint64_t* blind_in, const uint64_t* blind_tab, uint64_t* perm_in, uint64_t* perm_tab);
/* shplonk prover: out = sum_j coeffs[j] * polys[j] - low (low: nlow leading coefficients); a <- a / (X - root) */
void zko_linear_combination(const uint64_t* const* polys, size_t npolys, size_t n, const uint64_t* coeffs, const uint64_t* low, size_t nlow,
                            uint64_t* out);
void zko_kate_division(uint64_t* a, size_t n, const uint64_t root[4]);
void zko_eval_polynomials(const uint64_t* const* polys, size_t npolys, size_t n, const uint64_t x[4], uint64_t* out);

/* ---- synthetic data (repo-wide spec; also csrc/synth.hip) ---- */
uint64_t zko_splitmix64(uint64_t x);
void zko_synth_raw253(uint64_t seed, uint64_t idx, uint64_t out[4]);
void zko_synth_fill(uint64_t seed, uint64_t first, size_t n, uint64_t* out);   /* n x 4 */
/* fixed-base scalar multiples of the generator: out_xy[i] = [scalars[i]] G (scalars Montgomery Fr) */
void zko_fixed_base_mul(const uint64_t* scalars, size_t n, uint64_t* out_xy, int threads);
/* ParamsKZG::setup scalars (halo2_proofs poly/kzg/commitment.rs): monomial[i] = s^i,
 * lagrange[i] = l_i(s); n = 2^k; all Montgomery Fr. */
void zko_kzg_setup_scalars(uint32_t k, const uint64_t s[4], uint64_t* monomial, uint64_t* lagrange);
/* eval_polynomial (Horner) */
void zko_eval_polynomial(const uint64_t* coeffs, size_t n, const uint64_t x[4], uint64_t out[4]);

#ifdef __cplusplus
}
#endif
#endif
