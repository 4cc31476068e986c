/* zkhip.h — C ABI of libzkhip.so: the MI355X (gfx950) backend for the arithmetic underneath
 * halo2_proofs::plonk::create_proof (BN254 G1 MSM, Fr NTT, quotient sweep).
 *
 * The reference reaches this arithmetic through generic Rust calls with no FFI seam of its own:
 *   gen_snark_shplonk      /root/reference/src/helpers.rs:233,299; src/bin/cli.rs:320,343,369,462
 *   gen_evm_proof_shplonk  /root/reference/src/bin/cli.rs:519
 * -> halo2_proofs::plonk::create_proof (crate pinned at Cargo.lock:1320-1322)
 * -> halo2curves msm::best_multiexp / fft::best_fft (crate pinned at Cargo.lock:1359-1361).
 * Each entry point below names the upstream function whose body a [patch]ed crate would replace
 * with a call to it ([UPSTREAM-RECALL]: upstream sources are not on this machine); the Rust-side
 * bindings are in INTEGRATION.md.
 *
 * Conventions
 *  - Field elements: 4 little-endian u64 limbs, Montgomery form (R = 2^256) = the in-memory layout
 *    of halo2curves bn256::{Fr, Fq}.  G1Affine = {x, y} (8 u64, identity all-zero).  G1 = Jacobian
 *    {x, y, z} (12 u64, identity z = 0).
 *  - Every function returns 0 on success or a negative ZKHIP_E* code; zkhip_last_error() gives the
 *    message for the calling thread.  Nothing throws across the boundary.
 *  - Buffers are caller-owned.  "_device" entry points take device addresses (hipMalloc or
 *    torch.Tensor.data_ptr()) and are asynchronous on the context's stream; the others take host
 *    pointers and return when the result is in host memory.
 *  - One context per process and GPU; calls on one context must not race.
 */
#ifndef ZKHIP_H
#define ZKHIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZKHIP_OK 0
#define ZKHIP_EINVAL (-1)   /* bad argument */
#define ZKHIP_EHIP (-2)     /* HIP runtime / launch failure */
#define ZKHIP_ENOMEM (-3)
#define ZKHIP_EPROGRAM (-4) /* malformed evaluation graph */
#define ZKHIP_ENODEVICE (-5)
#define ZKHIP_ECONSTRAINT (-6) /* the witness violates the argument (halo2 Error::ConstraintSystemFailure) */

typedef struct zkhip_ctx zkhip_ctx;
typedef struct zkhip_srs zkhip_srs;
typedef struct zkhip_domain zkhip_domain;

/* ---- context ---- */
int  zkhip_init(zkhip_ctx** out, int device_id);
void zkhip_destroy(zkhip_ctx* ctx);
const char* zkhip_last_error(void);
/* Launch on an existing HIP stream (e.g. torch.cuda.current_stream().cuda_stream; NULL is HIP's
 * default stream, which is also torch's default).  ZKHIP_OWN_STREAM selects the context's own
 * non-blocking stream, the initial setting. */
#define ZKHIP_OWN_STREAM ((void*)(intptr_t)-1)
int  zkhip_set_stream(zkhip_ctx* ctx, void* hip_stream);
int  zkhip_synchronize(zkhip_ctx* ctx);
/* Tuning knobs (DESIGN.md lists them).  The ZKHIP_* environment variables are read once, in zkhip_init; this sets one afterwards
 * by its environment name ("ZKHIP_MSM_C") or short name ("msm_c").  Options that shape SRS tables (msm_c) apply to tables
 * built afterwards. */
int  zkhip_set_option(zkhip_ctx* ctx, const char* name, int value);
/* Host buffers handed to the library (zkhip_msm_g1 / _batch slices; zkhip_create_proof_ex's advice_on_host columns and host zk_blinding) may be plain pageable memory — a
 * Rust Vec<Fr>: they must stay alive and unmodified until the call returns, nothing more.  For large ones (>= 2^20 scalars; >= 64 MiB of advice in all) the library spawns a
 * short-lived WORKER THREAD per call that issues the host-to-device copies (a copy from pageable memory blocks its calling thread; this way it is not the thread that launches
 * the kernels) and joins it before returning, on every path; option "host_copy_thread" = 0 keeps everything on the calling thread, "host_register" = 1 additionally pins the
 * buffers with hipHostRegister for the call (slow for pages that were not pinned recently: off by default).  profiles/r06_host_inputs.txt has the numbers. */
/* Frees the scratch buffers of streams the context no longer uses (scratch is per stream).  Synchronises the device. */
int  zkhip_trim(zkhip_ctx* ctx);
/* Drops what the context keeps per proving key (zk_proving_key.key_id): the sorted lookup tables and the key's columns in the coset
 * layout.  Call it when a ProvingKey is dropped; a later proof with the same key_id rebuilds them.  Synchronises the device. */
int  zkhip_key_release(zkhip_ctx* ctx, uint64_t key_id);
int  zkhip_malloc(zkhip_ctx* ctx, size_t bytes, void** dptr);
int  zkhip_free(zkhip_ctx* ctx, void* dptr);
int  zkhip_memcpy_h2d(zkhip_ctx* ctx, void* dst, const void* src, size_t bytes);
int  zkhip_memcpy_d2h(zkhip_ctx* ctx, void* dst, const void* src, size_t bytes);
/* HIP-event pair on the context's stream, for timing the kernels where they actually run. */
int  zkhip_timer_start(zkhip_ctx* ctx);
int  zkhip_timer_stop_ms(zkhip_ctx* ctx, float* ms);   /* synchronises */

/* Per-kernel timing: while enabled, every launch of the named kernels is bracketed by HIP events on
 * the launch stream.  Names: "msm_digits", "msm_plan", "msm_accum_affine", "msm_accum_jac", "msm_tail",
 * "ntt_strided", "ntt_final", "sweep", "lookup_permute", "grand_product", "batch_invert", "eval_polynomial",
 * "linear_combination", "kate_division".  zkhip_profile_enable also clears the record. */
int  zkhip_profile_enable(zkhip_ctx* ctx, int on);
/* Restrict the recording to one kernel name (NULL = all): a proof issues ~60 timed spans, and their event records cost ~4 % of
 * a 10 ms proof, so a benchmark times only the kernel it reports live and takes the full breakdown in a separate pass. */
int  zkhip_profile_select(zkhip_ctx* ctx, const char* kernel);
int  zkhip_profile_read(zkhip_ctx* ctx, const char* kernel, double* total_ms, uint64_t* launches);
/* Work counters accumulated while profiling ALL kernels (no selection): "msm_pairs" = (non-zero digit, point) pairs the MSM
 * accumulations really processed (zero digits are skipped), "msm_dense_pairs" = n * windows per column.  Counted always, on a context with
 * a communicator: "proofs_row_sharded" (zkhip_create_proof_ex calls that exchanged row windows instead of complete columns),
 * "proofs_pieces_sharded" (... whose quotient pieces stayed row ranges), "shplonk_row_sharded" (zkhip_shplonk_open calls on row ranges),
 * "comm_bulk" (1: the communicator has its bulk companion, see below), "collectives_bulk" (exchanges issued on it), "comm_selfcheck" (what
 * zkhip_comm_init's self-checks ran and passed, bits: 1 tagged grouped send / recv exchange on the first communicator, 2 ncclCommSplit gave the
 * bulk communicator on every rank, 4 tagged exchange on it, 8 ncclCommCount / ncclCommUserRank of both agree with (nranks, rank), 16 the groups
 * held the pair (rank, rank): a forced self-check of a one-rank communicator) and "ctx_dead" (1: a host wait of this context gave up at
 * comm_timeout_ms; every later wait fails at once, zkhip_create_proof_ex / zkhip_msm_g1 / zkhip_msm_g1_batch are refused on entry with ZKHIP_EHIP, and
 * zkhip_destroy abandons the device-side resources instead of waiting for them). */
int  zkhip_profile_counter(zkhip_ctx* ctx, const char* name, uint64_t* value);

/* ---- one proof over several GPUs: one process per GPU, RCCL over xGMI ----
 * zkhip_comm_unique_id (one rank) produces the 128 opaque bytes of an RCCL unique id; the launcher distributes them
 * (torch.distributed, MPI, a file) and every rank calls zkhip_comm_init.  With a communicator on the context:
 *   - zkhip_kzg_setup_range / zkhip_srs_load_range give each rank the window tables of its point range only (1/N of the table);
 *   - every MSM entry point over such an SRS is COLLECTIVE: each rank sums its slice, the ncols x 96-byte partial sums are
 *     all-gathered and folded on the device, every rank receives the complete sums;
 *   - zkhip_create_proof_ex additionally distributes the coset NTTs by polynomial, keeps the cosets ROW-SHARDED (an all-to-all of row
 *     windows — own row range + halo of every coset block — instead of all-gathers of complete columns, whenever the quotient runs on
 *     cosets and 64 N divides n), sweeps by row range, shards the evaluations by query, and — with MSMs by point range — keeps the
 *     quotient's pieces, h(X) and SHPLONK's polynomials as row ranges (zkhip_shplonk_open does the same on its own when its SRS handle
 *     is this rank's point range: linear combinations on the range, divisions by X - r with the carries exchanged as 32-byte range
 *     totals).  Every rank must call it with identical inputs and obtains the identical proof; zk_proof_out.d_h then holds this
 *     rank's rows of the pieces only.  Option "row_sharded" = 0 restores the all-gather form.
 *   - the all-to-alls of row windows — tens of megabytes per peer that only the sweep reads — ride on a BULK communicator: a split of the
 *     first one over the same ranks (ncclCommSplit: no second id to distribute) with its own stream, created and self-checked by
 *     zkhip_comm_init when option "comm_bulk" = 1 (ZKHIP_COMM_BULK=1), so the latency-sized exchanges a commitment waits for never queue
 *     behind them.  OFF by default since round 6: two communicators' kernels must co-reside on every rank to make progress, which has only
 *     met stand-ins of librccl so far, and a caller without a fallback ladder should not be the first to try (bench.py opts in on its first
 *     rung).  A library without ncclCommSplit, a rank whose split / stream / buffer could not be made (agreed BEFORE anyone uses the split), or a
 *     failed self-check on any rank leave them on the first communicator (same results).
 *   - option "comm_selfcheck_force" = 1 (ZKHIP_COMM_SELFCHECK_FORCE): zkhip_comm_init on a ONE-rank communicator still runs every self-check
 *     with the pair (rank, rank) in each group and creates the bulk communicator — one process on one GPU drives ncclSend / ncclRecv /
 *     ncclGroupStart / ncclGroupEnd / ncclCommSplit / ncclCommCount / ncclCommUserRank and the destroy order (split before parent) of the
 *     real librccl; counter "comm_selfcheck" says what passed.  A bring-up check, not a data path.
 * zkhip_comm_init_host is the same with the all-gathers staged through host memory and a caller-supplied function (bring-up on a
 * one-GPU box, launchers without RCCL): fn(user, send, recv, bytes) must fill recv[r * bytes ..] with rank r's send block. */
typedef int (*zkhip_host_allgather_fn)(void* user, const void* send, void* recv, size_t bytes_per_rank);
/* the host transport's all-to-all (optional): send / recv hold nranks blocks of bytes_per_pair; block r of send goes to rank r, block r of
 * recv comes from rank r.  Without it the library emulates the exchange through the all-gather callback (N times the volume). */
typedef int (*zkhip_host_alltoall_fn)(void* user, const void* send, void* recv, size_t bytes_per_pair);
/* Which collective library the RCCL transport binds (process-wide; before the first zkhip_comm_unique_id / zkhip_comm_init): a path for
 * dlopen — another RCCL build than the one already in the process, or a stand-in with the same entry points (this repo's one-GPU tests
 * drive the transport's code path through tests/fake_rccl).  NULL or never called: the librccl the process already uses (torch ships
 * one), else the system's.  The library reads no environment variable for this. */
int  zkhip_comm_use_library(const char* path);
int  zkhip_comm_unique_id(uint8_t id[128]);
int  zkhip_comm_init(zkhip_ctx* ctx, const uint8_t id[128], int rank, int nranks);
int  zkhip_comm_init_host(zkhip_ctx* ctx, int rank, int nranks, zkhip_host_allgather_fn fn, void* user);
int  zkhip_comm_set_host_alltoall(zkhip_ctx* ctx, zkhip_host_alltoall_fn fn, void* user);
int  zkhip_comm_destroy(zkhip_ctx* ctx);
/* MSMs over WHOLE-SRS handles (every rank holds the full window tables) on a context with a communicator: on = 1 splits every batch by
 * column — rank r commits columns r, r + N, ... completely, results all-gathered — the split of choice while one MSM cannot fill
 * several GPUs (k <= 19: SURVEY.md 8(e)-2, the north star's "independent column commitments shard across the GPUs"); on = 0 (default)
 * every rank computes every column.  Point-range handles (zkhip_kzg_setup_range) are always split by point range. */
int  zkhip_comm_shard_columns(zkhip_ctx* ctx, int on);
int  zkhip_comm_info(const zkhip_ctx* ctx, int* rank, int* nranks, uint64_t* bytes_gathered);
/* transport <- "rccl" / "host" / "none"; transport_ranks <- the rank count the transport itself reports (ncclCommCount for RCCL, -1 if that
 * symbol is missing); collectives <- exchanges issued so far.  What a launcher prints to show that RCCL really spans N processes. */
int  zkhip_comm_describe(const zkhip_ctx* ctx, char* transport, size_t cap, int* transport_ranks, uint64_t* collectives);
int  zkhip_comm_allgather_device(zkhip_ctx* ctx, const void* d_send, void* d_recv, size_t bytes_per_rank);
/* Per-exchange trace of a context with an RCCL communicator (a measurement aid, off by default; bench.py --replay-rank records one rank's
 * exchange timeline with it).  zkhip_comm_trace(ctx, 1) clears the record and marks time zero on the context's stream (an event) and on the host
 * clock; from then on every all-gather / grouped send-recv exchange leaves an entry: the phase of the proof it belongs to (zkhip_comm_phase_name
 * of phase[i]: "advice", "lookup permute", "grand products", "quotient", "evaluations", "shplonk", "" outside a proof), the bytes this rank
 * receives, the host time of the issue and the time the exchange COMPLETED on the communicator's stream (a timing event right behind it), both
 * in microseconds since the mark; flags bit 0 = on the bulk communicator, bit 1 = send / recv group (else all-gather).  zkhip_comm_trace_read waits
 * for the context's stream, writes min(n, cap) entries (any array may be NULL) and end_us = when the context's stream drained. */
int  zkhip_comm_trace(zkhip_ctx* ctx, int on);
int  zkhip_comm_trace_read(zkhip_ctx* ctx, size_t cap, size_t* n, uint8_t* phase, uint64_t* bytes_received, double* host_issue_us, double* stream_done_us,
                           uint8_t* flags, double* end_us);
const char* zkhip_comm_phase_name(uint8_t id);

/* ---- SRS: ParamsKZG::{g, g_lagrange} (halo2_proofs src/poly/kzg/commitment.rs) ----
 * Uploaded once; the device keeps, per base, its W window multiples 2^(c*w) * P_i in affine form
 * so that an MSM is a single bucket pass with no doublings (sized for 288 GB of HBM). */
int  zkhip_srs_load(zkhip_ctx* ctx, const uint64_t* bases_xy, size_t n, zkhip_srs** out);
int  zkhip_srs_load_device(zkhip_ctx* ctx, const void* d_bases_xy, size_t n, zkhip_srs** out);
void zkhip_srs_free(zkhip_ctx* ctx, zkhip_srs* srs);
size_t zkhip_srs_len(const zkhip_srs* srs);
/* window width c (bits) and number of windows W = ceil(255 / c) chosen for this SRS: one MSM over n scalars
 * is n * W (digit, point) pairs. */
void zkhip_srs_window(const zkhip_srs* srs, uint32_t* c, uint32_t* windows);
/* ParamsKZG::setup(k, rng) restricted to G1: bases[i] = [s^i] G (monomial) and [l_i(s)] G
 * (Lagrange), generated on the device.  s is a Montgomery Fr.  Either output may be NULL. */
int  zkhip_kzg_setup(zkhip_ctx* ctx, uint32_t k, const uint64_t s[4], zkhip_srs** g, zkhip_srs** g_lagrange);
/* The same for the bases [first, first + count) only: one rank's share of a point-range-sharded SRS.  The handle remembers the
 * range; an MSM over it takes GLOBAL point indices and is collective when the context has a communicator. */
int  zkhip_kzg_setup_range(zkhip_ctx* ctx, uint32_t k, const uint64_t s[4], size_t first, size_t count, zkhip_srs** g, zkhip_srs** g_lagrange);
/* bases_xy: the `count` bases [first, first + count) of an SRS of n_total points (e.g. one rank's slice of a params file) */
int  zkhip_srs_load_range(zkhip_ctx* ctx, const uint64_t* bases_xy, size_t n_total, size_t first, size_t count, zkhip_srs** out);
/* global range [first, first + count) held by this handle (the whole SRS unless sharded) */
void zkhip_srs_range(const zkhip_srs* srs, size_t* first, size_t* count, size_t* n_total);
/* Copies base points [first, first+count) back (affine, Montgomery) — for tests. */
int  zkhip_srs_read(zkhip_ctx* ctx, const zkhip_srs* srs, size_t first, size_t count, uint64_t* out_xy);

/* ---- MSM: halo2curves msm::best_multiexp(coeffs, bases) -> G1, as called by
 * ParamsKZG::commit / commit_lagrange with bases = srs[..n] ----
 * zkhip_msm_g1: out_xyz is the sum in Jacobian form with z = 1 (or the identity (0,1,0)). */
int  zkhip_msm_g1(zkhip_ctx* ctx, const zkhip_srs* srs, const uint64_t* scalars, size_t n, uint64_t out_xyz[12]);
/* ncols HOST columns of n scalars in one call: out_xyz receives ncols x 12 u64, each sum normalised like zkhip_msm_g1's.  Both host forms
 * are pipelined from 2^20 scalars (round 6): the slices are registered with the runtime for the duration of the call (they must stay alive and unmoved until it
 * returns), uploaded on a copy stream chunk by chunk (option "msm_host_chunks") and summed as their bytes land; the batch form does not synchronise between its columns
 * (column j + 1 is on the wire while column j is summed).  Same results as the device-resident forms for every chunking. */
int  zkhip_msm_g1_batch(zkhip_ctx* ctx, const zkhip_srs* srs, const uint64_t* const* scalar_cols, size_t ncols, size_t n, uint64_t* out_xyz);
/* ncols independent columns in one pass; d_scalar_cols is a HOST array of device pointers,
 * d_out_xyz receives ncols x 12 u64 (device): Jacobian sums, any representative (normalise with
 * zkhip_g1_to_affine after fetching).  Asynchronous apart from one 4-byte-per-column read-back. */
int  zkhip_msm_g1_batch_device(zkhip_ctx* ctx, const zkhip_srs* srs, const void* const* d_scalar_cols,
                               size_t ncols, size_t n, void* d_out_xyz);
/* The same over the point range [first, first+count): scalar first+i of every column pairs with base
 * first+i.  One rank's share of a point-range-sharded MSM; the partial sums are
 * exchanged as raw bytes and folded with zkhip_g1_add. */
int  zkhip_msm_g1_batch_range_device(zkhip_ctx* ctx, const zkhip_srs* srs, const void* const* d_scalar_cols,
                                     size_t ncols, size_t first, size_t count, void* d_out_xyz);
/* The same with one SRS per column (all of the same length): lets commitments over g and g_lagrange that
 * do not depend on each other (advice columns + the vanishing argument's random polynomial) share one pass. */
int  zkhip_msm_g1_multi_device(zkhip_ctx* ctx, const zkhip_srs* const* srs_per_col, const void* const* d_scalar_cols,
                               size_t ncols, size_t first, size_t count, void* d_out_xyz);
/* G1 + G1 on the host (Jacobian, 12 u64 each). */
void zkhip_g1_add(const uint64_t a[12], const uint64_t b[12], uint64_t out[12]);
/* G1::to_affine on the host for results fetched from the device (12 u64 -> 8 u64). */
void zkhip_g1_to_affine(const uint64_t xyz[12], uint64_t out_xy[8]);
/* n points at once with a single field inversion (Curve::batch_normalize). */
void zkhip_g1_batch_to_affine(const uint64_t* xyz, size_t n, uint64_t* out_xy);
/* G1Affine::to_bytes (32-byte compressed) */
void zkhip_g1_to_bytes(const uint64_t xy[8], uint8_t out[32]);
/* What create_proof does with a batch of commitments: the n Jacobian results of zkhip_msm_g1_*_device (device memory) ->
 * host, batch_normalize, and (if out_bytes != NULL) the 32-byte form transcript.write_point consumes.  One stream sync. */
int  zkhip_commitments_read(zkhip_ctx* ctx, const void* d_xyz, size_t n, uint64_t* out_xy, uint8_t* out_bytes);

/* ---- NTT: halo2curves fft::best_fft(a, omega, log_n): in place, natural order in and out ---- */
int  zkhip_fft(zkhip_ctx* ctx, uint64_t* a, const uint64_t omega[4], uint32_t log_n);
/* batch of npolys arrays; d_polys is a HOST array of device pointers.  Asynchronous. */
int  zkhip_fft_batch_device(zkhip_ctx* ctx, void* const* d_polys, size_t npolys, const uint64_t omega[4], uint32_t log_n);

/* ---- EvaluationDomain (halo2_proofs src/poly/domain.rs) ----
 * zkhip_domain_new(j, k) = EvaluationDomain::new(j, k); g_coset NULL = Fr::ZETA. */
int  zkhip_domain_new(zkhip_ctx* ctx, uint32_t j, uint32_t k, const uint64_t g_coset[4], zkhip_domain** out);
void zkhip_domain_free(zkhip_ctx* ctx, zkhip_domain* dom);
uint32_t zkhip_domain_k(const zkhip_domain* dom);
uint32_t zkhip_domain_extended_k(const zkhip_domain* dom);
uint32_t zkhip_domain_quotient_poly_degree(const zkhip_domain* dom);
void zkhip_domain_constants(const zkhip_domain* dom, uint64_t omega[4], uint64_t extended_omega[4], uint64_t g_coset[4]);
/* lagrange_to_coeff / coeff_to_lagrange: npolys arrays of n, in place. */
int  zkhip_lagrange_to_coeff_device(zkhip_ctx* ctx, const zkhip_domain* dom, void* const* d_polys, size_t npolys);
int  zkhip_coeff_to_lagrange_device(zkhip_ctx* ctx, const zkhip_domain* dom, void* const* d_polys, size_t npolys);
/* coeff_to_extended: in[i] has n_in coefficients (zero-extended), out[i] has extended_n values. */
int  zkhip_coeff_to_extended_device(zkhip_ctx* ctx, const zkhip_domain* dom, const void* const* d_in, size_t n_in,
                                    void* const* d_out, size_t npolys);
/* extended_to_coeff: extended_n values in place; the first n*quotient_poly_degree are the result. */
int  zkhip_extended_to_coeff_device(zkhip_ctx* ctx, const zkhip_domain* dom, void* const* d_polys, size_t npolys);
/* divide_by_vanishing_poly: a[i] *= t_evaluations[i mod 2^(extended_k-k)], in place. */
int  zkhip_divide_by_vanishing_device(zkhip_ctx* ctx, const zkhip_domain* dom, void* d_a);
/* ---- the same quotient on quotient_poly_degree cosets of the size-n domain (cosets.hip) ----
 * The extended domain g<w_ext> is the union of E = 2^(extended_k - k) cosets s_r<w>, s_r = g w_ext^r; h has fewer than q n coefficients
 * (q = quotient_poly_degree), so q of them determine it.  When q < E (cs.degree() = 4: 3 of 4; 6: 5 of 8) zkhip_create_proof_ex works
 * on the first q cosets: fewer transform and sweep rows, the same h, pieces, commitments and proof bytes as EvaluationDomain's
 * coeff_to_extended / divide_by_vanishing_poly / extended_to_coeff whenever the numerator is a multiple of X^n - 1, i.e. for every
 * witness that satisfies the circuit (for one that does not, neither result is a quotient and the two differ; zkhip_set_option(ctx,
 * "coset_quotient", 0) keeps the extended domain).  A column in this layout is q blocks of n values, block r = the polynomial on
 * s_r<w> in natural order = extended index r + E i.  These entry points expose the pieces for tests and for callers with their own
 * schedule; all return ZKHIP_EINVAL when q >= E. */
int  zkhip_domain_cosets(zkhip_ctx* ctx, const zkhip_domain* dom, uint32_t* q, uint64_t* shifts /* q x 4 (ABI form) or NULL */);
/* d_in[i]: n coefficients; d_out[i]: q n values */
int  zkhip_coeff_to_cosets_device(zkhip_ctx* ctx, const zkhip_domain* dom, const void* const* d_in, void* const* d_out, size_t npolys);
/* d_vals: q n numerator values (overwritten); d_pieces: q n coefficients of numerator / (X^n - 1), piece j at element j n */
int  zkhip_cosets_to_pieces_device(zkhip_ctx* ctx, const zkhip_domain* dom, void* d_vals, void* d_pieces);
/* Host-pointer forms (upload, transform, download). */
int  zkhip_lagrange_to_coeff(zkhip_ctx* ctx, const zkhip_domain* dom, uint64_t* a);
int  zkhip_coeff_to_extended(zkhip_ctx* ctx, const zkhip_domain* dom, const uint64_t* coeffs, size_t n_in, uint64_t* out);
int  zkhip_extended_to_coeff(zkhip_ctx* ctx, const zkhip_domain* dom, uint64_t* a);

/* ---- quotient sweep: halo2_proofs plonk::evaluation::Evaluator::evaluate_h ----
 * Same flattened form the reference's Evaluator holds after keygen.
 * ValueSource = 3 x int32 {kind, a, b}: */
enum { ZK_VS_CONSTANT = 0, ZK_VS_INTERMEDIATE = 1, ZK_VS_FIXED = 2, ZK_VS_ADVICE = 3, ZK_VS_INSTANCE = 4,
       ZK_VS_CHALLENGE = 5, ZK_VS_BETA = 6, ZK_VS_GAMMA = 7, ZK_VS_THETA = 8, ZK_VS_Y = 9,
       ZK_VS_PREVIOUS = 10 };
/* Calculation record in the int32 code stream: {op, target, nsrc, nsrc x ValueSource}.
 * ADD/SUB/MUL: (a, b); SQUARE/DOUBLE/NEGATE/STORE: (a); HORNER: (start, factor, parts...). */
enum { ZK_OP_ADD = 0, ZK_OP_SUB = 1, ZK_OP_MUL = 2, ZK_OP_SQUARE = 3, ZK_OP_DOUBLE = 4, ZK_OP_NEGATE = 5,
       ZK_OP_HORNER = 6, ZK_OP_STORE = 7 };

typedef struct {
    const uint64_t* constants;   /* HOST: n_constants x 4, Montgomery */
    const int32_t*  rotations;   /* HOST: n_rotations */
    const int32_t*  code;        /* HOST: n_code_words int32 */
    uint32_t n_constants, n_rotations, n_code_words, n_calculations, n_intermediates;
} zk_graph;

typedef struct {
    uint32_t k, extended_k, cs_degree, blinding_factors;
    uint64_t extended_omega[4], g_coset[4], delta[4];
    uint64_t beta[4], gamma[4], theta[4], y[4];
    uint32_t n_fixed, n_advice, n_instance, n_challenges;
    const uint64_t* const* fixed_cosets;     /* HOST arrays of DEVICE pointers, each extended_n x 4 */
    const uint64_t* const* advice_cosets;
    const uint64_t* const* instance_cosets;
    const uint64_t* challenges;              /* HOST: n_challenges x 4 */
    const uint64_t* l0; const uint64_t* l_last; const uint64_t* l_active_row;   /* DEVICE */
    zk_graph custom_gates;
    uint32_t n_perm_columns, n_perm_sets;
    const uint32_t* perm_column_type;        /* HOST: 0 advice, 1 fixed, 2 instance */
    const uint32_t* perm_column_index;       /* HOST */
    const uint64_t* const* perm_sigma_cosets;    /* HOST array of DEVICE pointers, n_perm_columns */
    const uint64_t* const* perm_product_cosets;  /* n_perm_sets */
    uint32_t n_lookups, _pad;
    const zk_graph* lookup_graphs;                   /* HOST, n_lookups */
    const uint64_t* const* lookup_product_cosets;    /* HOST arrays of DEVICE pointers */
    const uint64_t* const* lookup_input_cosets;      /* permuted input A' */
    const uint64_t* const* lookup_table_cosets;      /* permuted table S' */
} zk_evalh_args;

/* d_out: extended_n x 4 u64 (device).  Asynchronous. */
int  zkhip_evaluate_h_device(zkhip_ctx* ctx, const zk_evalh_args* args, void* d_out);
/* the same for the extended rows [first_row, first_row + n_rows) only (n_rows a multiple of 64; d_out holds n_rows elements): the
 * row-sharded sweep of a proof spread over several GPUs — the columns are complete on every GPU, rotations wrap as usual */
int  zkhip_evaluate_h_rows_device(zkhip_ctx* ctx, const zk_evalh_args* args, size_t first_row, size_t n_rows, void* d_out);
/* The sweep over the coset layout above: every *_cosets pointer of `args` (and l0, l_last, l_active_row) is a q-block column, rows
 * [first_row, first_row + n_rows) of the q n; row r n + i is the natural sweep's extended row r + E i.  args->extended_omega / g_coset
 * are not used. */
int  zkhip_evaluate_h_cosets_device(zkhip_ctx* ctx, const zkhip_domain* dom, const zk_evalh_args* args, size_t first_row, size_t n_rows,
                                    void* d_out);

/* ---- grand products and evaluations: the O(n) field work of create_proof between the commitments ----
 * All columns are DEVICE arrays of n = 2^k ABI field elements; pointer lists are HOST arrays.  Asynchronous.
 * ff::BatchInvert: a[i] <- 1 / a[i], zeros stay zero, in place. */
int  zkhip_batch_invert_device(zkhip_ctx* ctx, void* d_a, size_t n);
/* arithmetic::eval_polynomial for a batch: d_out[j] = polys[j](x), npolys x 4 u64 on the device. */
int  zkhip_eval_polynomial_device(zkhip_ctx* ctx, const void* const* d_polys, size_t npolys, size_t n, const uint64_t x[4], void* d_out);
/* The same with one point per polynomial (xs: npolys x 4 u64 on the HOST): every (polynomial, x * omega^rotation) query of
 * create_proof in one pass. */
int  zkhip_eval_polynomials_at_device(zkhip_ctx* ctx, const void* const* d_polys, size_t npolys, size_t n, const uint64_t* xs, void* d_out);
/* plonk::permutation::prover::Argument::commit, up to the commitment: the grand-product polynomials z (Lagrange form), one
 * per chunk of chunk_len = cs.degree() - 2 columns.  values[j] / sigmas[j]: the j-th permutation column and its sigma
 * polynomial, both in Lagrange form.  The last blinding_factors rows of every z are copied from d_blinding
 * ([set][blinding_factors] elements: upstream draws them from its rng). */
int  zkhip_permutation_products_device(zkhip_ctx* ctx, uint32_t k, const void* const* d_values, const void* const* d_sigmas, size_t ncols,
                                       uint32_t chunk_len, const uint64_t beta[4], const uint64_t gamma[4], uint32_t blinding_factors,
                                       const void* d_blinding, void* const* d_z);
/* plonk::lookup::prover::permute_expression_pair (halo2_proofs plonk/lookup/prover.rs): A' = the theta-compressed input
 * column sorted over the usable rows n - (blinding_factors + 1); S'[i] = A'[i] where A'[i] != A'[i-1], the unused table
 * values (ascending) fill the other rows (descending); the last blinding_factors + 1 rows of both take d_blind_in /
 * d_blind_tab ([blinding_factors + 1] elements each, drawn by the caller).  Synchronises the stream.  Returns
 * ZKHIP_ECONSTRAINT if an input value does not occur in the table. */
int  zkhip_permute_expression_pair_device(zkhip_ctx* ctx, uint32_t k, uint32_t blinding_factors, const void* d_input, const void* d_table,
                                          const void* d_blind_in, const void* d_blind_tab, void* d_perm_in, void* d_perm_tab);
/* plonk::lookup::prover::Permuted::commit_product: z of one lookup from the theta-compressed input / table columns and
 * their permuted forms. */
int  zkhip_lookup_product_device(zkhip_ctx* ctx, uint32_t k, const void* d_compressed_input, const void* d_compressed_table,
                                 const void* d_permuted_input, const void* d_permuted_table, const uint64_t beta[4],
                                 const uint64_t gamma[4], uint32_t blinding_factors, const void* d_blinding, void* d_z);

/* Both of the above for a whole proof behind a single batch inversion (an inversion pass is latency-bound: 380 dependent
 * products).  Either part may be empty (ncols = 0 / n_lookups = 0).  Lookup arrays are HOST arrays of device columns;
 * d_lookup_blinding holds [lookup][blinding_factors] elements. */
int  zkhip_grand_products_device(zkhip_ctx* ctx, uint32_t k, const uint64_t beta[4], const uint64_t gamma[4], uint32_t blinding_factors,
                                 const void* const* d_values, const void* const* d_sigmas, size_t ncols, uint32_t chunk_len,
                                 const void* d_perm_blinding, void* const* d_perm_z,
                                 size_t n_lookups, const void* const* d_compressed_input, const void* const* d_compressed_table,
                                 const void* const* d_permuted_input, const void* const* d_permuted_table, const void* d_lookup_blinding,
                                 void* const* d_lookup_z);

/* ---- SHPLONK multi-open prover arithmetic (halo2_proofs poly/kzg/multiopen/shplonk/prover.rs, reached through
 * gen_snark_shplonk at /root/reference/src/helpers.rs:233,299) ----
 * d_out[i] = sum_j coeffs[j] * d_polys[j][i] - (i < nlow ? low[i] : 0): the y- / v-power combinations of the rotation sets'
 * polynomials minus the low-degree interpolant.  d_polys is a HOST array of device polynomials of n coefficients; coeffs
 * (npolys x 4) and low (nlow x 4, nlow <= n) are host arrays; d_out must not alias an input. */
int  zkhip_linear_combination_device(zkhip_ctx* ctx, size_t n, const void* const* d_polys, size_t npolys, const uint64_t* coeffs,
                                     const uint64_t* low, size_t nlow, void* d_out);
/* d_dst[j] = d_src[j] / (X - roots[j]) (remainder dropped, top coefficient zero) for npolys polynomials of n coefficients in
 * one pass; d_dst[j] may equal d_src[j].  The shplonk prover divides an exact multiple N of prod_t (X - r_t) through
 * partial fractions: N / prod_t (X - r_t) = sum_t N / (X - r_t) / prod_{s != t} (r_t - r_s), i.e. independent divisions. */
int  zkhip_divide_by_linear_device(zkhip_ctx* ctx, size_t n, const void* const* d_src, void* const* d_dst, size_t npolys, const uint64_t* roots);
/* arithmetic::kate_division, in place and batched: polynomial j (n coefficients) is divided by prod_t (X - roots[j][t]) over
 * its nroots[j] roots (roots: host, all polynomials' roots concatenated); remainders are dropped and the vacated top
 * coefficients are zero, i.e. the result is already "resized to n". */
int  zkhip_kate_division_device(zkhip_ctx* ctx, size_t n, void* const* d_polys, size_t npolys, const uint32_t* nroots, const uint64_t* roots);

/* ProverSHPLONK::create_proof (poly/kzg/multiopen/shplonk/prover.rs) in one call, with the transcript as callbacks in the
 * order upstream uses it: squeeze y, squeeze v, write_point(h), squeeze u, write_point(h').  Query i opens polynomial
 * d_polys[query_poly[i]] (n coefficients, device) at query_points[i] with value query_evals[i] (host arrays, nq x 4 u64, ABI form;
 * the caller has computed the evaluations, e.g. with zkhip_eval_polynomials_at_device).  Rotation sets are formed as in
 * construct_intermediate_sets (commitments in query order, points ascending).  Scalars cross the callbacks in ABI form, points
 * as 32 compressed bytes and as affine (x, y).  Any number of distinct points / rotation-set sizes (upstream has no limit). */
typedef struct zk_transcript {
    void* user;
    void (*write_point)(void* user, const uint8_t bytes32[32], const uint64_t xy[8]);
    void (*squeeze_challenge)(void* user, uint64_t out[4]);
    void (*write_scalar)(void* user, const uint64_t scalar[4]);   /* evaluations (zkhip_create_proof only; may be NULL for shplonk_open) */
    /* Transcript::common_scalar: absorbed, not written to the proof (the verifying key's transcript_repr and the instance values at the
     * start of create_proof).  May be NULL: then nothing is absorbed there (a caller that has already done it on its own transcript). */
    void (*common_scalar)(void* user, const uint64_t scalar[4]);
} zk_transcript;
int  zkhip_shplonk_open(zkhip_ctx* ctx, const zkhip_srs* srs, size_t n, const void* const* d_polys, size_t npolys, const uint32_t* query_poly,
                        const uint64_t* query_points, const uint64_t* query_evals, size_t nq, const zk_transcript* transcript,
                        uint64_t h1_xy[8], uint64_t h2_xy[8]);

/* ---- halo2_proofs::plonk::create_proof in one call (plonk/prover.rs; reached from gen_snark_shplonk at
 * /root/reference/src/helpers.rs:233,299 and src/bin/cli.rs:320,343,369,462), from the point where the witness columns exist.
 * zk_proving_key is what keygen leaves on the device: every array is a HOST array of DEVICE columns unless noted.
 * Transcript order (upstream's plonk/prover.rs [UPSTREAM-RECALL]): common_scalar(vk.transcript_repr), common_scalar(every instance
 * value), advice points, squeeze theta, per lookup its permuted input and permuted table points, squeeze beta, gamma, product
 * points (permutation sets, then lookups), the random-polynomial point, squeeze y, quotient piece points, squeeze x, then the
 * evaluations in the order upstream WRITES them: advice queries, fixed queries, random polynomial, sigma polynomials, per
 * permutation set z(x), z(wx) and (all but the last set) z(w^last x), per lookup z(x), z(wx), a'(x), a'(w^-1 x), s'(x) —
 * h(x) is NOT written — then SHPLONK (squeeze y', v', point h, squeeze u, point h').  The multi-open itself takes its queries in
 * upstream's QUERY order (advice, permutation products, lookups, fixed, sigma, h, random), which is a different order. */
typedef struct zk_proving_key {
    uint32_t k, cs_degree, blinding_factors;
    uint32_t n_fixed, n_advice, n_instance, n_lookups, n_perm_columns;
    const zkhip_srs* g;
    const zkhip_srs* g_lagrange;
    const zkhip_domain* domain;
    const void* const* fixed_lagrange; const void* const* fixed_coeff; const void* const* fixed_cosets;   /* n_fixed */
    const void* const* sigma_lagrange; const void* const* sigma_coeff; const void* const* sigma_cosets;   /* n_perm_columns */
    const void* l0; const void* l_last; const void* l_active_row;                                         /* DEVICE, extended */
    zk_graph custom_gates;
    const zk_graph* lookup_graphs;            /* HOST, n_lookups: the lookup argument's input/table product graphs (evaluate_h) */
    const zk_graph* lookup_input_compress;    /* HOST, n_lookups: theta-compression of the input expressions */
    const zk_graph* lookup_table_compress;    /* HOST, n_lookups */
    /* Optional shortcuts (HOST arrays of n_lookups entries, or NULL): when lookup i's input is exactly one advice column at rotation 0,
     * lookup_input_advice_column[i] is its index (else -1) and no compression pass runs; when its table is exactly one FIXED column at
     * rotation 0 (halo2-lib's range table), lookup_table_fixed_column[i] is its index (else -1): no compression pass, and — if key_id
     * is non-zero — the column is sorted once and the sorted form kept by the context under (key_id, i) instead of being sorted in
     * every proof.  key_id must then be unique per proving key (fixed-column contents) for the life of the context. */
    const int32_t* lookup_input_advice_column;
    const int32_t* lookup_table_fixed_column;
    uint64_t key_id;
    const uint32_t* perm_column_type;         /* HOST: 0 advice, 1 fixed, 2 instance */
    const uint32_t* perm_column_index;
    uint32_t n_advice_queries, n_fixed_queries;                                     /* cs.advice_queries / cs.fixed_queries order */
    const uint32_t* advice_query_column; const int32_t* advice_query_rotation;      /* HOST */
    const uint32_t* fixed_query_column; const int32_t* fixed_query_rotation;        /* HOST */
    uint64_t delta[4];                                                              /* Fr::DELTA, ABI form */
    const uint64_t* vk_transcript_repr;       /* HOST, 4 u64 (ABI) or NULL: pk.vk.transcript_repr, absorbed first (vk.hash_into) */
    /* Advice phases and user challenges (cs.advice_column_phase / cs.challenge_phase [UPSTREAM-RECALL: axiom's create_proof loops over
     * the phases: commit the advice columns of phase p, write them, squeeze the challenges whose phase is p]).  NULL / 0: every column
     * in phase 0, no user challenge — the reference's three circuits. */
    const uint8_t* advice_column_phase;       /* HOST, n_advice, or NULL */
    uint32_t n_challenges;
    const uint8_t* challenge_phase;           /* HOST, n_challenges (non-decreasing use is not required: challenge j is squeezed after phase challenge_phase[j]).
                                                 Checked (ZKHIP_EINVAL): the advice phases are 0 .. max without a gap, no challenge's phase exceeds the last
                                                 advice phase, and zk_proof_inputs.advice_phase is set whenever a phase > 0 exists */
} zk_proving_key;
typedef struct zk_proof_out {
    const void* d_h;          /* the quotient in coefficient form (quotient_poly_degree * n elements, library-owned, valid until the next proof) */
    uint64_t* evals;          /* caller's HOST buffer, evals_cap x 4 (may be NULL): every opened evaluation, h's included, in QUERY order */
    uint32_t* eval_poly;      /* caller's HOST buffers, evals_cap each (may be NULL): polynomial table index and rotation per evaluation */
    int32_t* eval_rotation;
    uint32_t* eval_write_order; /* caller's HOST buffer, evals_cap (may be NULL): indices into the above in the order the evaluations were
                                   written to the transcript (n_evals - 1 entries: h(x) is not written) */
    size_t evals_cap, n_evals;
} zk_proof_out;
/* What upstream draws from the caller's rng inside create_proof, in the caller's hands (plonk/prover.rs passes `rng` down to
 * lookup::commit_permuted, permutation::commit, lookup::commit_product and vanishing::commit [UPSTREAM-RECALL]; reached from
 * gen_snark_shplonk at /root/reference/src/helpers.rs:233 with the sdk's rng).  All four buffers are field elements in ABI form;
 * either all DEVICE or all HOST (on_host = 1: uploaded by the library).  The blinding rows of the ADVICE columns are already part of
 * the advice columns the caller hands over (the last blinding_factors + 1 rows of each). */
typedef struct zk_blinding {
    const void* lookup_permuted;   /* [lookup][2][blinding_factors + 1]: permuted input rows, then permuted table rows */
    const void* perm_z;            /* [permutation set][blinding_factors] */
    const void* lookup_z;          /* [lookup][blinding_factors] */
    const void* random_poly;       /* the vanishing argument's random polynomial: n coefficients */
    int on_host;
} zk_blinding;
/* Witness synthesis of a later advice phase, in the caller's hands (upstream runs the circuit's `synthesize` again with the challenges
 * of the earlier phases): called once per phase p > 0, after the challenges of phases < p are known.  challenges: HOST, n_challenges x 4
 * u64 (ABI; entries of later phases are zero).  advice: the SAME array zk_proof_inputs.advice points to — the callee stores the pointers
 * of phase p's columns into it (DEVICE or HOST as advice_on_host says; the buffers must stay valid until zkhip_create_proof_ex returns).
 * Return 0, or non-zero to abort the proof (ZKHIP_EINVAL is returned). */
typedef int (*zk_advice_phase_fn)(void* user, uint32_t phase, const uint64_t* challenges, const void** advice);
typedef struct zk_proof_inputs {
    const void* const* advice;                /* n_advice columns of n elements, Lagrange form (columns of later phases: see advice_phase) */
    int advice_on_host;                       /* 0: DEVICE columns (resident pipeline); 1: HOST columns (the Vec<Fr> a Rust caller holds;
                                                 uploaded by the library, fastest from pinned memory) */
    const void* const* d_instance;            /* n_instance DEVICE columns (zero-padded to n), or NULL: built from instance_values */
    const uint64_t* const* instance_values;   /* HOST, per column instance_len[i] x 4 u64: absorbed with common_scalar in upstream's order
                                                 (KZG: instances are hashed, not committed); NULL: nothing absorbed */
    const uint32_t* instance_len;             /* HOST, n_instance */
    const zk_blinding* blinding;              /* NULL: the library's counter generator seeded with blinding_seed
                                                 (zkhip_synth_fill_device seeds +300.., +320.., +340, +360, +380) */
    uint64_t blinding_seed;
    zk_advice_phase_fn advice_phase;          /* required when the key has a column of phase > 0, else ignored (may be NULL) */
    void* advice_phase_user;
} zk_proof_inputs;
/* A zk_blinding that is given must be complete: every member whose row count is non-zero for this key (lookup_permuted / lookup_z with
 * lookups, perm_z with permutation sets, random_poly always) must be non-NULL, else ZKHIP_EINVAL — never a silent fall-back to the
 * seeded generator.  All argument checks happen before the first asynchronous copy from the caller's memory; on an error after that
 * point the call waits for its copies before it returns, so the caller may drop its buffers.
 * PHASES [UPSTREAM-RECALL: axiom's create_proof commits the advice columns phase by phase and squeezes the user challenges of a phase
 * after its commitments]: implemented since round 4 — zk_proving_key.advice_column_phase / challenge_phase say which column and which
 * challenge belongs to which phase, zk_proof_inputs.advice_phase supplies the witness of the later phases, the challenges reach the gate
 * and lookup expressions as ZK_VS_CHALLENGE.  The three circuits of the reference (RSA, zkevm SHA-256, the aggregation circuit;
 * /root/reference/src/bin/cli.rs:296-527) use phase 0 only: for them nothing changes (all three fields NULL / 0). */
int  zkhip_create_proof_ex(zkhip_ctx* ctx, const zk_proving_key* pk, const zk_proof_inputs* in, const zk_transcript* transcript,
                           zk_proof_out* out);
/* 1 if zkhip_create_proof_ex will evaluate this key's quotient on quotient_poly_degree cosets of the size-n domain — the key's
 * extended-domain forms (fixed_cosets, sigma_cosets, l0, l_last, l_active_row) may then be NULL — 0 if it will work on the extended
 * domain and needs them (option "coset_quotient" off, key_id 0, degree too high, coefficient forms missing).  The predicate the prover
 * itself evaluates: a binding asks it instead of re-deriving the rule. */
int  zkhip_coset_quotient_applies(const zkhip_ctx* ctx, const zk_proving_key* pk);
/* The resident-pipeline short form: device columns, seeded blinding, no instance absorption. */
int  zkhip_create_proof(zkhip_ctx* ctx, const zk_proving_key* pk, const void* const* d_advice, const void* const* d_instance,
                        uint64_t blinding_seed, const zk_transcript* transcript, zk_proof_out* out);

/* ---- a ready-made transcript: halo2_proofs transcript.rs Blake2bWrite<Vec<u8>, G1Affine, Challenge255<_>> on the host
 * (BLAKE2b-512 personalised "Halo2-Transcript"; point = prefix 1 + canonical x, y; scalar = prefix 2 + canonical bytes; challenge =
 * prefix 0, digest of a state copy, reduced mod r).  Its writer's bytes — compressed points and scalars in transcript order — are
 * the proof.  A Rust caller passes callbacks into its own transcript instead. */
typedef struct zkhip_blake2b_transcript zkhip_blake2b_transcript;
zkhip_blake2b_transcript* zkhip_blake2b_transcript_new(void);
void zkhip_blake2b_transcript_free(zkhip_blake2b_transcript* t);
const zk_transcript* zkhip_blake2b_transcript_callbacks(zkhip_blake2b_transcript* t);
size_t zkhip_blake2b_transcript_proof(const zkhip_blake2b_transcript* t, const uint8_t** bytes);          /* -> length */
size_t zkhip_blake2b_transcript_points(const zkhip_blake2b_transcript* t, const uint64_t** xy);           /* -> count; 8 u64 each */
size_t zkhip_blake2b_transcript_challenges(const zkhip_blake2b_transcript* t, const uint64_t** limbs);    /* -> count; 4 u64 each (ABI) */

/* ---- and snark-verifier's EvmTranscript (system/halo2/transcript/evm.rs; behind gen_evm_proof_shplonk, /root/reference/src/bin/cli.rs:519):
 * Keccak-256 over a byte buffer of big-endian coordinates / scalars; the proof stream holds points as 64 bytes (x, y big-endian) and
 * scalars as 32 big-endian bytes.  The identity point cannot be written (upstream returns an error; here its (0, 0) is written). */
typedef struct zkhip_evm_transcript zkhip_evm_transcript;
zkhip_evm_transcript* zkhip_evm_transcript_new(void);
void zkhip_evm_transcript_free(zkhip_evm_transcript* t);
const zk_transcript* zkhip_evm_transcript_callbacks(zkhip_evm_transcript* t);
size_t zkhip_evm_transcript_proof(const zkhip_evm_transcript* t, const uint8_t** bytes);          /* -> length */
size_t zkhip_evm_transcript_challenges(const zkhip_evm_transcript* t, const uint64_t** limbs);    /* -> count; 4 u64 each (ABI) */
size_t zkhip_evm_transcript_points(const zkhip_evm_transcript* t, const uint64_t** xy);           /* -> count; 8 u64 each */
void zkhip_keccak256(const uint8_t* in, size_t len, uint8_t pad /* 0x01 Keccak-256, 0x06 SHA3-256 */, uint8_t out[32]);

/* ---- and snark-verifier's PoseidonTranscript<G1Affine, NativeLoader, _, T = 3, RATE = 2, R_F = 8, R_P = 57> (system/halo2/transcript/
 * halo2.rs over util/hash/poseidon.rs) — the transcript behind gen_snark_shplonk, i.e. behind every leaf proof and the aggregation
 * snark of the reference (/root/reference/src/helpers.rs:233,299; src/bin/cli.rs:320,343,369,462).  The permutation's parameters come
 * from the Grain LFSR generator of the Poseidon reference and reproduce its published poseidonperm_x5_254_3 vector; the sponge layer
 * (initial state 2^64, buffering, padding, squeeze = state[1]) is restated from recall.  Proof stream: 32-byte compressed points,
 * 32-byte little-endian scalars. */
typedef struct zkhip_poseidon_transcript zkhip_poseidon_transcript;
zkhip_poseidon_transcript* zkhip_poseidon_transcript_new(void);
void zkhip_poseidon_transcript_free(zkhip_poseidon_transcript* t);
const zk_transcript* zkhip_poseidon_transcript_callbacks(zkhip_poseidon_transcript* t);
size_t zkhip_poseidon_transcript_proof(const zkhip_poseidon_transcript* t, const uint8_t** bytes);          /* -> length */
size_t zkhip_poseidon_transcript_points(const zkhip_poseidon_transcript* t, const uint64_t** xy);           /* -> count; 8 u64 each */
size_t zkhip_poseidon_transcript_challenges(const zkhip_poseidon_transcript* t, const uint64_t** limbs);    /* -> count; 4 u64 each (ABI) */
void zkhip_poseidon_permute(uint64_t state[12]);               /* the bare permutation, 3 ABI elements in place (partial rounds in sparse form) */
void zkhip_poseidon_permute_plain(uint64_t state[12]);         /* the same by textbook rounds (dense MDS product in every round) */
void zkhip_poseidon_params(uint64_t* rc, uint64_t* mds);       /* 65 x 3 round constants, 3 x 3 MDS (row-major), ABI form; either may be NULL */

/* ---- synthetic tables (bench / tests): element i of a column = raw253(seed, i) taken as the
 * Montgomery limbs (oracle/pyref.py synth_raw253) ---- */
int  zkhip_synth_fill_device(zkhip_ctx* ctx, void* d_out, size_t n, uint64_t seed, uint64_t first_index);
/* small-valued column (the witness-shaped distributions of SURVEY.md 8(d): bit / word columns): with h = splitmix64(seed +
 * (first_index + i) * 0x2545F4914F6CDD1D), element i is the integer (h >> 32) & 1 if (h mod 2^32) mod 1000 < bits_per_mille,
 * else (h >> 32) mod 2^word_bits (1 <= word_bits <= 32), stored in Montgomery form */
int  zkhip_synth_small_device(zkhip_ctx* ctx, void* d_out, size_t n, uint64_t seed, uint64_t first_index, uint32_t bits_per_mille,
                              uint32_t word_bits);

#ifdef __cplusplus
}
#endif
#endif
