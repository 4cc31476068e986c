// common.hpp — context, error plumbing and scratch memory shared by the libzkhip.so translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <deque>
#include <chrono>
#include <map>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../include/zkhip.h"
#include "bn254.hpp"

namespace zk {

void set_error(const char* fmt, ...);

#define ZK_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            zk::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return ZKHIP_EHIP;                                                                    \
        }                                                                                         \
    } while (0)

#define ZK_TRY(expr)                \
    do {                            \
        int _rc = (expr);           \
        if (_rc != ZKHIP_OK) return _rc; \
    } while (0)

#define ZK_LAUNCH_CHECK() ZK_HIP(hipGetLastError())

// every device allocation of the library goes through here: the time spent inside hipMalloc is what a cold process pays before its
// first proof (58 GB at k = 22), and zkhip_profile_counter("alloc_us" / "alloc_bytes" / "alloc_calls") reports it
struct AllocStats { std::atomic<uint64_t> ns{0}, bytes{0}, calls{0}; };   // process-wide, any thread
extern AllocStats g_alloc;   // ctx.hip
hipError_t dev_malloc(void** p, size_t bytes);

// Named scratch buffers owned by the context, grown on demand and reused across calls so the hot
// path never calls hipMalloc (MI355X has 288 GB: scratch is sized for the largest call seen).
struct Scratch {
    void* ptr = nullptr;
    size_t bytes = 0;
};

}  // namespace zk

// Tuning knobs: environment variables (ZKHIP_*) read ONCE when the context is created, or set afterwards with
// zkhip_set_option — never looked up on the hot path.  0 / -1 = "use the measured default".
struct zkhip_options {
    int debug_delay_main_us = 0;   // test aid, the reverse: the MAIN stream is held this long right after it has issued a side-stream section
    int debug_delay_us = 0;    // test aid: every side-stream / third-stream section of a proof starts with a kernel that holds that stream this long (prover.hip k_debug_delay)
    int host_copy_thread = 1;  // zkhip_create_proof_ex issues its large host uploads (advice_on_host, a host random polynomial) from a worker thread: pageable sources do not hold the proof's thread (0: from the proof's thread)
    int host_register = 0;     // ... and / or registers those buffers with the runtime for the call (pays ~0.7 ms per buffer for pages not pinned recently: off by default)
    int msm_host_chunks = 0;   // zkhip_msm_g1 (host slice): pieces the upload + MSM pipeline is cut into (0: by size — 4 from 2^21 scalars, 2 from 2^20, else 1)
    int msm_c = 0, msm_seg = 0, msm_tailparts = 0, msm_ch = 0, msm_widetail = -1, msm_tail2 = -1, msm_adaptive_l = 1, msm_debug = 0;
    int sort_hb = 0, sort_tile = 0, sort_one_atomic = 1, sort_copies = 0, sort_wide = -1;   // sort_wide: low-pass block shape (-1: 1024 threads x 8 pairs for 8192-pair tiles)
    int ntt_lds_pad = 0;   // analysis only: KiB of unused dynamic LDS added to every register-tiled NTT workgroup (fewer tiles per CU: the occupancy-vs-time curve)
    int ntt_smax = 0, ntt_r8 = 4, ntt_group = 0;   // ntt_r8: 0 stage-per-barrier, 1 8 per thread, 2 / 3 4 per thread on 2048 / 1024 tiles, 4 auto
    int permute_rank_sort = 1, eval_byval = 1, late_overlap = -1;
    int host_timing = 0;   // zkhip_create_proof prints its host-side phase times to stderr
    int row_sharded = 1;      // multi-rank proofs on the coset path: all-to-all of row windows (1) / all-gather of complete columns (0)
    int comm_bulk = 0;        // multi-rank proofs: 1 = the all-to-alls of row windows (needed by the sweep only) ride on a SECOND communicator (ncclCommSplit) with its own stream, so the
                              // latency-sized exchanges a commitment waits for never queue behind a 50 MB transfer; 0 (the library's default since round 6): everything on the one
                              // communicator.  Two communicators' kernels must co-reside on every rank to make progress and that has only met stand-ins, so callers opt in
                              // (bench.py's first rung does, with a fallback ladder behind it; a Rust caller has no ladder)
    int comm_selfcheck_force = 0;   // zkhip_comm_init on a ONE-rank communicator still runs the self-checks, with the pair (rank, rank) in every group, and creates the bulk communicator (comm.hip)
    int comm_timeout_ms = 120000;   // host waits of a multi-rank context give up after this long (0: wait for ever): see zk::CommWatch
    int eval_chunks = -1;     // evaluations at x in this many launches, absorbed chunk by chunk while the next is computed (1: one launch; -1: by size)
    int rand_overlap = -1;    // the vanishing argument's random polynomial is committed on a third stream beside the grand products (0: with the advice batch; -1: by size)
    int coset_quotient = 1;   // zkhip_create_proof evaluates the quotient on quotient_poly_degree cosets of size n (cosets.hip) when that is fewer rows
};

// Multi-GPU state of a context (comm.hip): rank / size, the RCCL communicator with its own stream (or the host-staged transport),
// and a byte counter for the exchange volume.
struct zkhip_comm {
    int rank = 0, nranks = 1;
    void* nccl = nullptr;
    hipStream_t stream = nullptr;             // every collective is enqueued here, fenced against the producing / consuming stream
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    // the BULK communicator (round 5): a split of `nccl` over the same ranks with its own stream and events; carries the all-to-alls of row
    // windows (prover.hip to_extended) and nothing else.  null: those ride on `nccl` like everything else (no ncclCommSplit in the library, the
    // option comm_bulk = 0, or its self-check failed on some rank)
    void* nccl_bulk = nullptr;
    hipStream_t stream_bulk = nullptr;
    hipEvent_t ev_in_bulk = nullptr, ev_out_bulk = nullptr;
    uint64_t collectives_bulk = 0;            // of `collectives`: issued on the bulk communicator
    zkhip_host_allgather_fn host_allgather = nullptr;
    void* host_user = nullptr;
    zkhip_host_alltoall_fn host_alltoall = nullptr;   // optional companion of the host transport (else emulated through the all-gather)
    void* host_alltoall_user = nullptr;
    void* stage = nullptr;        // pinned staging buffer of the host transport
    size_t stage_bytes = 0;
    uint64_t bytes_gathered = 0;  // bytes this rank received through RCCL all-gathers
    uint64_t collectives = 0;     // exchanges issued so far
    int shard_columns = 0;        // MSMs over whole-SRS handles: 1 = split the batch by column over the ranks, 0 = replicate
    int a2a_ok = 0;               // verdict of zkhip_comm_init's all-to-all self-check: 1 passed on every rank, -1 failed somewhere, 0 not run
    // Per-exchange trace (zkhip_comm_trace: a measurement aid, off by default; RCCL branch only): for every exchange the phase of the proof, the bytes this
    // rank receives, the host clock at the issue and a timing event recorded on the communicator's stream right behind the exchange.  Read back by
    // zkhip_comm_trace_read as (host_us, done_us) relative to the mark zkhip_comm_trace(ctx, 1) set on the context's stream.
    struct TraceEntry { const char* phase; uint64_t bytes; double host_us; hipEvent_t done; uint8_t flags; };
    bool trace_on = false;
    std::vector<TraceEntry> trace;
    std::vector<hipEvent_t> trace_pool;
    hipEvent_t trace_base = nullptr;
    std::chrono::steady_clock::time_point trace_t0;
    uint32_t selfcheck = 0;       // what zkhip_comm_init's self-checks ran and passed (bits: comm.hip SC_*; zkhip_profile_counter "comm_selfcheck")
    const char* phase = "";       // which part of the proof the host is issuing (what a timed-out wait reports)
    mutable int stuck = 0;        // a host wait of this context ran into comm_timeout_ms: the communicator's stream (and whatever waits on it) is taken for
                                  // dead from here on — later waits fail at once, error exits do not wait again, zkhip_comm_destroy does not touch RCCL
    // the exchange a multi-rank proof uses: row windows by grouped send / recv only where the user asked for it AND this communicator
    // has shown it can do it (the user's option itself is never overwritten)
    bool row_sharded(const zkhip_options& o) const { return o.row_sharded != 0 && a2a_ok >= 0; }
};

// Host -> device copies of the caller's PAGEABLE buffers, issued from a thread of their own.  hipMemcpyAsync from pageable memory blocks the calling host thread for the
// copy's duration (only registered / pinned sources are asynchronous), and registering is no general answer: pinning pages that were not pinned recently costs ~0.7 ms per
// hipHostRegister CALL (a fresh Vec<Fr> per proof is always cold; 32 columns: +22 ms — profiles/r06_host_inputs.txt).  So zkhip_create_proof_ex hands its large uploads to this
// worker: the copies block THAT thread, the proof's own thread goes on launching, and before it makes a stream wait for a job's event it waits (host side) until the worker
// has recorded it.  Jobs run in order on one stream; done = jobs finished so far; joined before the call returns, on every path.
// The copy stream the jobs go to depends on nothing but an event recorded on the main stream when the call began, so the worker cannot end up behind a collective of THIS call; a
// context whose earlier call was given up on (zkhip_ctx::dead) is refused before a worker is started.  What is not covered: a main stream already stuck behind a collective of an
// EARLIER call that no host wait has noticed yet — the worker's first pageable copy then blocks as the same copy issued from the caller's thread always did, without a deadline.
struct zk_copy_job { void* dst; const void* src; size_t bytes; hipEvent_t ev; };
struct zk_copy_worker {
    std::vector<zk_copy_job> jobs;
    std::thread th;
    std::atomic<size_t> done{0};
    std::atomic<int> err{0};
    // -> false: no thread could be created (the jobs are then run by the caller, in order: run_inline) — no exception crosses the C ABI
    bool start(int device, hipStream_t stream) {
        try {
        th = std::thread([this, device, stream] {
            hipError_t e = hipSetDevice(device);
            for (auto& j : jobs) {
                if (e == hipSuccess && j.bytes) e = hipMemcpyAsync(j.dst, j.src, j.bytes, hipMemcpyHostToDevice, stream);
                if (e == hipSuccess && j.ev) e = hipEventRecord(j.ev, stream);
                if (e != hipSuccess) err.store((int)e, std::memory_order_relaxed);
                done.fetch_add(1, std::memory_order_release);
            }
        });
        } catch (...) { return false; }
        return true;
    }
    hipError_t run_inline(hipStream_t stream) {
        for (auto& j : jobs) {
            hipError_t e = hipSuccess;
            if (j.bytes) e = hipMemcpyAsync(j.dst, j.src, j.bytes, hipMemcpyHostToDevice, stream);
            if (e == hipSuccess && j.ev) e = hipEventRecord(j.ev, stream);
            if (e != hipSuccess) return e;
            done.fetch_add(1, std::memory_order_release);
        }
        return hipSuccess;
    }
    hipError_t wait(size_t job) {      // until job `job` (0-based) has been issued and its event recorded
        while (done.load(std::memory_order_acquire) <= job) std::this_thread::yield();
        return (hipError_t)err.load(std::memory_order_relaxed);
    }
    void join() { if (th.joinable()) th.join(); }
    ~zk_copy_worker() { join(); }
};

struct zkhip_ctx {
    int device = 0;
    zkhip_options opt;
    zkhip_comm comm;
    // A host wait of this context ran into comm_timeout_ms (set together with zkhip_comm::stuck, but it SURVIVES zkhip_comm_destroy, which resets `comm`):
    // some stream of the context is fenced behind a collective that will never complete.  Every later wait fails at once; zkhip_destroy neither
    // synchronises nor frees device memory nor destroys streams (each of those would block for ever): it abandons them and returns.
    mutable int dead = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    uint32_t max_seq = 0;           // tag of the last MSM's polled read-back (msm.hip)
    hipEvent_t ev_read = nullptr;   // marks a small device->host read in the middle of a launch sequence (see event_wait)
    hipEvent_t accum_mark = nullptr;     // if set, the next MSM records it right after its bucket-accumulation launch (and clears it)
    hipStream_t side_stream = nullptr;   // zkhip_create_proof's second stream (coset NTTs beside the MSM phases), created on first use
    hipEvent_t side_event = nullptr;
    hipStream_t aux_stream = nullptr;    // zkhip_create_proof's third stream: the random polynomial's commitment beside the (memory-bound) grand products
    hipEvent_t aux_event[2] = {nullptr, nullptr};
    hipEvent_t eval_event[4] = {nullptr, nullptr, nullptr, nullptr};   // one per chunk of the pipelined evaluations (zkhip_create_proof_ex)
    hipStream_t copy_stream = nullptr;   // uploads of large host advice columns (zkhip_create_proof_ex, advice_on_host), created on first use
    hipEvent_t copy_event[4] = {nullptr, nullptr, nullptr, nullptr};   // one per upload group
    hipEvent_t copy_event_rand = nullptr;        // the caller's random polynomial has landed (zkhip_create_proof_ex, host blinding + split upload)
    std::vector<hipEvent_t> host_chunk_event;   // zkhip_msm_g1's pipelined upload: one per chunk + one fence (created on first use)
    // Small host->device uploads of host temporaries (pointer tables, lowered programs): the bytes are copied into a pinned ring
    // and the asynchronous copy reads from there, so the call neither blocks on the stream nor keeps the caller's buffer alive.
    // The ring is 8 MiB against ~100 KiB staged per proof; wrapping around synchronises the device.
    void* stage_ring = nullptr;
    size_t stage_off = 0;
    static constexpr size_t STAGE_BYTES = 8u << 20;
    int upload(void* d_dst, const void* src, size_t bytes);
    void* h_pinned = nullptr;   // 64 KiB of pinned host memory for the small device->host reads on the critical path
    static constexpr size_t PINNED_BYTES = 64 * 1024;
    std::map<std::string, zk::Scratch> scratch;
    std::string scratch_tag;   // appended to every scratch name while set: zkhip_msm_g1's pipelined chunks keep their buffers apart (msm.hip)
    // name -> a buffer of ANOTHER phase that is dead while this name is in use (zkhip_create_proof_ex lends the advice cosets' block to
    // SHPLONK's quotient scratch): get_scratch hands it out instead of allocating, if it is large enough.  Set and cleared by the lender.
    std::map<std::string, zk::Scratch> lent;
    std::map<std::string, std::shared_ptr<void>> host_objects;   // host-side companions of persistent buffers (cosets.hip's plans)
    std::map<std::string, void*> persistent;   // named device buffers that outlive a call (keygen-like derived data), freed with the context
    // Twiddle tables keyed by (log_n, omega).  w^e for any e < 2^log_n is lo[e & (2^h - 1)] * hi[e >> h]
    // (two tables of ~sqrt(n) entries: L2-resident, against a 16 * n-byte table that every strided NTT pass
    // would gather from at random); bf[j] = w^(j * n / 2048) are the butterfly twiddles of one tile.
    struct Twiddle {
        uint32_t log_n;
        uint64_t omega[4];
        uint32_t h;          // bits of the low table
        void* d_lo;          // 2^h entries: w^j
        void* d_hi;          // 2^(log_n - h) entries: w^(j 2^h)
        void* d_bf;          // min(1024, n/2) entries: w^(j * n / 2^bf_bits'), see ntt.hip
        uint32_t bf_bits;    // bf has 2^bf_bits entries, bf[j] = w^(j << (log_n - 1 - bf_bits))
    };
    std::deque<Twiddle> twiddles;   // a deque: the pointers handed out stay valid however many tables are cached

    // optional per-kernel HIP-event timing on the launch stream (zkhip_profile_*)
    bool prof_on = false;
    std::string prof_only;   // if non-empty, only spans of this name are recorded (keeps the event traffic off the timed path)
    struct ProfSpan { const char* name; hipEvent_t e0, e1; };
    std::vector<ProfSpan> prof_spans;
    uint64_t n_row_sharded = 0, n_pieces_sharded = 0, n_shplonk_sharded = 0;   // multi-rank proofs by exchange mode (zkhip_profile_counter)
    uint64_t prof_msm_pairs = 0, prof_msm_dense_pairs = 0;   // (digit, point) pairs accumulated / n W per column, while profiling all kernels
    std::vector<hipEvent_t> prof_pool;
    hipEvent_t prof_event();
    void prof_begin(const char* name);
    void prof_end();

    int get_scratch(const char* name, size_t bytes, void** out);
    int get_twiddles(const uint64_t omega[4], uint32_t log_n, const Twiddle** out);
};

// RAII span around one kernel launch (no-op unless profiling is enabled)
struct ProfScope {
    zkhip_ctx* c;
    bool active;
    ProfScope(zkhip_ctx* ctx, const char* name) : c(ctx), active(ctx->prof_on && (ctx->prof_only.empty() || ctx->prof_only == name)) {
        if (active) c->prof_begin(name);
    }
    ~ProfScope() { if (active) c->prof_end(); }
};

// Host waits.  hipStreamSynchronize sleeps on an interrupt and wakes tens of microseconds late, and the prover has a dozen Fiat-Shamir
// round trips per proof on its critical path: the two waits below POLL (hipStreamQuery / hipEventQuery), reading the clock every 64 polls.
// A context with a communicator waits for its peers whenever it waits for its own stream — a collective whose partner never arrives (a
// rank that died, an RCCL that cannot connect two GPUs) would leave the host polling for ever — so while the context has a communicator
// (nranks > 1) the poll has a deadline (zkhip_options::comm_timeout_ms): on expiry the wait fails with hipErrorLaunchTimeOut and the error text names the rank,
// the number of collectives issued so far and the phase of the proof.  The deadline is the TOTAL duration of one host wait (not "no progress for
// N ms": the host cannot see progress inside a stream) and it applies to every wait of such a context — set it above the longest legitimate wait
// (N ranks time-slicing one device, a large queued batch).  After an expiry the context is marked stuck (zkhip_comm::stuck): every later wait fails
// at once.  Without a communicator: poll for 10 s, then the blocking wait.
namespace zk {
hipError_t wait_poll(const zkhip_ctx* c, hipStream_t st, hipEvent_t ev);   // ctx.hip: ev != null: wait for the event; else for the stream (null = the legacy default stream)
}
static inline hipError_t stream_wait(const zkhip_ctx* c, hipStream_t st) { return zk::wait_poll(c, st, nullptr); }
// Waits for an event recorded right after a small read-back, not for the whole stream: the kernels issued after the event keep running
// while the host acts on the value, so the next launches queue up behind them without a bubble.
static inline hipError_t event_wait(const zkhip_ctx* c, hipEvent_t ev) { return ev ? zk::wait_poll(c, nullptr, ev) : hipErrorInvalidResourceHandle; }

struct zkhip_domain;
namespace zk {
int comm_allgather(zkhip_ctx* ctx, const void* d_send, void* d_recv, size_t bytes);
int comm_allgather_begin(zkhip_ctx* ctx, const void* d_send, void* d_recv, size_t bytes);
int comm_allgather_end(zkhip_ctx* ctx);
int comm_alltoall(zkhip_ctx* ctx, const void* d_send, void* d_recv, size_t bytes_per_pair, const uint8_t* send_to = nullptr, const uint8_t* recv_from = nullptr,
                  bool bulk = false, bool with_self = false);   // bulk: on the bulk communicator when the context has one (same result either way)
struct RowCopy { const uint32_t* src; uint32_t* dst; uint32_t src_row0, dst_row0, count, src_mask, dst_mask; };
#define ZK_ROWCOPY_MAX 48
int comm_row_copies(zkhip_ctx* ctx, const std::vector<RowCopy>& list);
int comm_fold_partials(zkhip_ctx* ctx, const void* d_part, size_t ncols, void* d_out);
// msm.hip: one HOST column through the chunk pipeline of zkhip_msm_g1, split in two for zkhip_create_proof_ex (uploads issued early, the commitment later)
size_t host_column_chunks(const zkhip_ctx* ctx, const zkhip_srs* srs, size_t n);
int host_column_jobs(zkhip_ctx* ctx, const void* host, size_t n, size_t K, void* d_col, std::vector<zk_copy_job>* jobs);   // the K chunk copies (each with its event) appended to a worker's list
int host_column_commit(zkhip_ctx* ctx, const zkhip_srs* srs, size_t n, size_t K, const void* d_col, void* d_out, zk_copy_worker* worker, size_t first_job);
int lagrange_to_coeff_oop(zkhip_ctx* ctx, const zkhip_domain* d, const void* const* srcs, void* const* dsts, size_t npolys);
int permute_expression_pair_async(zkhip_ctx* ctx, uint32_t k, uint32_t blinding_factors, const void* d_input, const void* d_table,
                                  const void* d_blind_in, const void* d_blind_tab, void* d_perm_in, void* d_perm_tab, uint32_t* d_err_flag,
                                  const void* d_sorted_table_keys);
int permute_sorted_table_keys(zkhip_ctx* ctx, uint64_t key_id, uint32_t slot, uint32_t k, uint32_t blinding_factors, const void* d_table,
                              const void** d_keys);
int ntt_tabled(zkhip_ctx* ctx, const void* const* srcs, void* const* dsts, size_t npolys, const uint64_t omega[4], uint32_t log_n,
               const void* d_pre_tab, const void* d_post_tab, uint32_t tab_group, size_t tab_stride);
int power_table(zkhip_ctx* ctx, const uint64_t base_abi[4], size_t count, void* d_out);
// The quotient sweep over q cosets s_r H of the size-2^k domain H instead of the extended domain (cosets.hip): every column is q
// blocks of n values (block r = the polynomial on s_r H, natural order), a rotation stays inside its block.
struct SweepCosets { uint32_t q; const uint64_t* shifts_abi; /* q x 4: s_r */ const uint64_t* omega_abi; /* the size-n root */ };
int evaluate_h_cosets(zkhip_ctx* ctx, const zk_evalh_args* A, const SweepCosets* cs, size_t first_row, size_t n_rows, void* d_out);
// cosets.hip
struct CosetPlan;
int coset_plan(zkhip_ctx* ctx, const zkhip_domain* d, const CosetPlan** out);
uint32_t coset_plan_q(const CosetPlan* p);
int coeff_to_cosets(zkhip_ctx* ctx, const CosetPlan* p, const void* const* srcs, void* const* dsts, size_t npolys);
int cosets_to_pieces(zkhip_ctx* ctx, const CosetPlan* p, void* d_vals, void* d_pieces);
int grand_products_range(zkhip_ctx* ctx, uint32_t k, const uint64_t beta[4], const uint64_t gamma[4], uint32_t blinding_factors,
                         const void* const* d_values, const void* const* d_sigmas, size_t ncols, uint32_t chunk_len, const void* d_perm_blinding,
                         void* const* d_perm_z, size_t n_lookups, const void* const* d_compressed_input, const void* const* d_compressed_table,
                         const void* const* d_permuted_input, const void* const* d_permuted_table, const void* d_lookup_blinding,
                         void* const* d_lookup_z, size_t row0, size_t count);   // polyops.hip: rows [row0, row0 + count) of every z (collective when count < n)
int cosets_inverse_blocks(zkhip_ctx* ctx, const CosetPlan* p, void* d_vals, const uint32_t* blocks, size_t nblocks);
int cosets_combine_range(zkhip_ctx* ctx, const CosetPlan* p, const void* d_vals, void* d_pieces, size_t first_row, size_t count);
struct KeyCosets { std::vector<const void*> fixed, sigma; const void* l0; const void* l_last; const void* l_active; };
int key_cosets(zkhip_ctx* ctx, const CosetPlan* p, const zk_proving_key* pk, const KeyCosets** out);
void coset_sweep_view(const CosetPlan* p, SweepCosets* out);
int shplonk_open(zkhip_ctx* ctx, const zkhip_srs* srs, size_t n, const void* const* d_polys, size_t npolys, const uint32_t* query_poly,
                 const uint64_t* query_points, const uint64_t* query_evals, size_t nq, const zk_transcript* tr, uint64_t h1_xy[8], uint64_t h2_xy[8],
                 bool rows_only);   // shplonk.hip: zkhip_shplonk_open; rows_only = the polynomials exist as this rank's row range only
void coset_forget_key(zkhip_ctx* ctx, uint64_t key_id);   // the plans' host-side entries of a released key (the device buffers are the caller's to free)
}
static inline unsigned div_up(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }
