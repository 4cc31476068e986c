// prover.hip — halo2_proofs::plonk::create_proof as one library call over the kernels of this library.
//
// Mirrors the order of operations of plonk/prover.rs create_proof [UPSTREAM-RECALL; crate pinned at
// /root/reference/Cargo.lock:1320-1322; reached from gen_snark_shplonk at /root/reference/src/helpers.rs:233,299 and
// src/bin/cli.rs:320,343,369,462] from the point where the witness columns exist (witness synthesis is the circuit's job):
//   advice commitments -> theta -> lookup compression + permute_expression_pair + commitments -> beta, gamma ->
//   permutation / lookup grand products + commitments -> random polynomial commitment -> y -> quotient (coset NTTs, sweep,
//   division, split) + commitments -> x -> evaluations -> SHPLONK multi-open.
// The transcript stays with the caller: commitments, evaluations, absorbed-only scalars (vk, instances) and challenges cross four
// callbacks, in upstream's order; the evaluations are WRITTEN in upstream's order and opened in upstream's (different) query order.
// Inputs (zk_proof_inputs): device or host advice columns, instance values, the caller's rng draws.  On a context with a communicator
// (comm.hip) the same call runs one proof over several GPUs.
// Everything here is host orchestration (no kernels): the point of having it in the library is that a proof of 2^17 rows is
// ~9 ms of GPU work, and an interpreted host adds a millisecond of gaps between ~250 launches.
// halo2-zkcert_amd/prover.py is the same schedule in Python over the small entry points (and the form the oracle backend runs).
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <thread>
#include <vector>

#include "common.hpp"
#include "hostfield.hpp"
using namespace zk;

namespace {
// Work issued between begin() and end() runs on a second stream, ordered after everything already issued on the main one
// (the coset NTTs of finished columns beside the latency-bound MSM phases); join() makes the main stream wait for it.
// Debug aid (option debug_delay_us / ZKHIP_DEBUG_DELAY_US, off by default): every section of work that the proof puts on a stream OTHER than its main one — the side stream's
// overlapped transforms, the third stream's random-polynomial commitment, the sharded proof's communicator streams are not touched — starts with a kernel that holds that stream for
// this many microseconds.  The proof's bytes must not notice: a consumer on the main stream that lacks its event dependency on such a section (or a later writer of a buffer the
// section still reads) is invisible at normal timing, shows up once in a while when other processes take turns on the device, and shows up ALWAYS with the section held back.
__global__ void k_debug_delay(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
static void debug_delay(zkhip_ctx* ctx, hipStream_t s, int us = -1) {
    if (us < 0) us = ctx->opt.debug_delay_us;
    if (us <= 0) return;
    static int khz = [] { int v = 100000; (void)hipDeviceGetAttribute(&v, hipDeviceAttributeWallClockRate, 0); return v; }();
    hipLaunchKernelGGL(k_debug_delay, dim3(1), dim3(1), 0, s, (unsigned long long)((double)us * 1e-3 * khz));
    (void)hipGetLastError();
}
struct Overlap {
    zkhip_ctx* ctx;
    hipStream_t main, side;
    hipEvent_t ev;
    int begin() {
        ZK_HIP(hipEventRecord(ev, main));
        ZK_HIP(hipStreamWaitEvent(side, ev, 0));
        if (side != main) debug_delay(ctx, side);
        ctx->stream = side;
        return ZKHIP_OK;
    }
    // The same, but the side stream starts when the MSM launched since arm() has finished its bucket accumulation (the MSM records
    // the event there): the overlapped NTTs then fill the MSM's latency-bound tail, the host round trip and the start of the next
    // phase instead of competing with the accumulation for the whole chip.
    // Measured: at 2^17 this is worth 2 % of the proof and lifts the accumulation from 0.54 to 0.60 of the mad peak; at 2^19 / 2^22,
    // where tails are negligible and the accumulation leaves issue slots free, starting the NTTs at once is 3-9 % better (late = false).
    bool late;
    void arm() { ctx->accum_mark = late ? ev : nullptr; if (!late) (void)hipEventRecord(ev, main); }
    int begin_marked() {
        if (late && ctx->accum_mark) { ctx->accum_mark = nullptr; ZK_HIP(hipEventRecord(ev, main)); }   // no MSM consumed it: mark now
        ZK_HIP(hipStreamWaitEvent(side, ev, 0));
        if (side != main) debug_delay(ctx, side);
        ctx->stream = side;
        return ZKHIP_OK;
    }
    void end() { ctx->stream = main; if (side != main && ctx->opt.debug_delay_main_us > 0) debug_delay(ctx, main, ctx->opt.debug_delay_main_us); }      // (the reverse experiment: the MAIN stream held back behind every section it has just issued)
    int join() {
        ZK_HIP(hipEventRecord(ev, side));
        ZK_HIP(hipStreamWaitEvent(main, ev, 0));
        return ZKHIP_OK;
    }
};
struct StreamGuard {   // whatever happens, the context leaves on its main stream
    zkhip_ctx* ctx;
    hipStream_t main;
    bool host_uploads = false, aux_launched = false, done = false;
    // Option host_register (off by default): the caller's large host columns are registered with the runtime for the duration of the call.  A Rust Vec<Fr> is PAGEABLE
    // memory, and an unregistered source makes every hipMemcpyAsync block the CALLING THREAD for the copy's duration; registered, the copies are asynchronous — but pinning
    // pages that were not pinned recently costs ~0.7 ms per hipHostRegister call (a fresh Vec<Fr> per proof always is cold: 32 columns, +22 ms; the 2 us of
    // profiles/r06_h2d_probe.txt is the re-registration of warm pages), so the default is the upload thread instead (zk_copy_worker) and this stays for callers that keep
    // their columns in one long-lived allocation.  Fails harmlessly on memory that already is pinned.  Unregistered once the copies are known to be over.
    std::vector<const void*> registered;
    void pin(const void* h, size_t bytes) {
        if (std::find(registered.begin(), registered.end(), h) != registered.end()) return;
        if (hipHostRegister((void*)h, bytes, hipHostRegisterDefault) == hipSuccess) registered.push_back(h); else (void)hipGetLastError();
    }
    void unpin() {
        for (const void* h : registered) if (hipHostUnregister((void*)h) != hipSuccess) (void)hipGetLastError();
        registered.clear();
    }
    // Error exits.  (1) Asynchronous copies FROM THE CALLER'S host buffers may be in flight: wait for them, or a caller that drops its Vec<Fr> /
    // pinned buffers on the error races the DMA.  (2) The random polynomial's MSM may still be running on the third stream (rand_late): it reads
    // w_rand and writes the pinned commitment slot, which the NEXT proof reuses on the main stream — wait for it too.
    // The waits are stream_wait's (polling, with the communicator's deadline), never a bare hipStreamSynchronize.  When the error IS an expired
    // deadline (zkhip_comm::stuck / zkhip_ctx::dead) the streams fenced behind the communicator will never drain and are not waited for — but the
    // COPY stream never depends on the communicator (it waits only for an event recorded on the main stream before this proof's first collective), so
    // the caller's large host columns are still waited for, with a bounded poll of its own (bounded_drain: the sticky stuck flag would make stream_wait
    // return at once).  Small uploads ride on the main stream ahead of every collective of this proof: they completed before the collective that hung
    // could start.  What is left after a stuck exit: the main / side / aux streams; the caller's buffers are not read by them.
    static void bounded_drain(hipStream_t s, double limit_ms) {
        const auto t0 = std::chrono::steady_clock::now();
        while (hipStreamQuery(s) == hipErrorNotReady) {
            (void)hipGetLastError();
            if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > limit_ms) {
                fprintf(stderr, "zkhip_create_proof: the upload stream did not drain within %.0f ms of a stuck exit: keep the host columns alive until the process leaves\n", limit_ms);
                return;
            }
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
    }
    ~StreamGuard() {
        ctx->stream = main;
        if (done) { unpin(); return; }      // (a finished proof has read every column: its commitments and evaluations are in the transcript)
        if (ctx->comm.stuck || ctx->dead) {
            if (host_uploads && ctx->copy_stream) bounded_drain(ctx->copy_stream, 5000.0);
            if (!ctx->copy_stream || hipStreamQuery(ctx->copy_stream) == hipSuccess) unpin(); else (void)hipGetLastError();      // still in flight: the registration is abandoned with the stream
            return;
        }
        if (aux_launched && ctx->aux_stream) (void)stream_wait(ctx, ctx->aux_stream);
        if (host_uploads && !(ctx->comm.stuck || ctx->dead)) {
            if (ctx->copy_stream) (void)stream_wait(ctx, ctx->copy_stream);
            if (!(ctx->comm.stuck || ctx->dead)) (void)stream_wait(ctx, main);
        }
        if (!(ctx->comm.stuck || ctx->dead)) unpin();
    }
};
inline void abi_of(const HF& a, uint64_t out[4]) { fe32 m = hf_abi(a); memcpy(out, m.w, 32); }
}  // namespace

extern "C" int zkhip_create_proof(zkhip_ctx* ctx, const zk_proving_key* pk, const void* const* d_advice, const void* const* d_instance,
                                  uint64_t blinding_seed, const zk_transcript* tr, zk_proof_out* out) {
    zk_proof_inputs in;
    memset(&in, 0, sizeof in);
    in.advice = d_advice;
    in.d_instance = d_instance;
    in.blinding_seed = blinding_seed;
    if (pk && pk->n_instance && !d_instance) { set_error("zkhip_create_proof: null argument"); return ZKHIP_EINVAL; }
    return zkhip_create_proof_ex(ctx, pk, &in, tr, out);
}

extern "C" int zkhip_create_proof_ex(zkhip_ctx* ctx, const zk_proving_key* pk, const zk_proof_inputs* in, const zk_transcript* tr,
                                     zk_proof_out* out) {
    if (!ctx || !pk || !in || !tr || !tr->write_point || !tr->squeeze_challenge || !tr->write_scalar || (pk->n_advice && !in->advice) ||
        (pk->n_instance && !in->d_instance && !(in->instance_values && in->instance_len))) {
        set_error("zkhip_create_proof: null argument");
        return ZKHIP_EINVAL;
    }
    if (ctx->dead || ctx->comm.stuck) {      // a host wait of this context gave up on a collective earlier: its streams will never drain, nothing may be queued behind them (and no upload thread started)
        set_error("zkhip_create_proof: the context was given up on by an earlier host wait (comm_timeout_ms): only zkhip_destroy is left to call");
        return ZKHIP_EHIP;
    }
    const uint64_t blinding_seed = in->blinding_seed;
    const zk_blinding* bl = in->blinding;
    // host-side phase clock (ZKHIP_HOST_TIMING=1): where the host spends the time between the launches of a proof
    struct Mark { const char* what; std::chrono::steady_clock::time_point t; };
    std::vector<Mark> marks;
    const bool timing = ctx->opt.host_timing != 0;
    auto mark = [&](const char* what) { if (timing) marks.push_back({what, std::chrono::steady_clock::now()}); };
    mark("start");
    ctx->comm.phase = "advice";
    if (!pk->g || !pk->g_lagrange || !pk->domain) { set_error("zkhip_create_proof: proving key without SRS / domain"); return ZKHIP_EINVAL; }
    const uint32_t k = pk->k, bf = pk->blinding_factors;
    const size_t n = (size_t)1 << k;
    const uint32_t ek = zkhip_domain_extended_k(pk->domain), qd = zkhip_domain_quotient_poly_degree(pk->domain);
    const size_t en = (size_t)1 << ek;
    const uint32_t A = pk->n_advice, I = pk->n_instance, F = pk->n_fixed, L = pk->n_lookups, P = pk->n_perm_columns;
    if (pk->cs_degree < 3) { set_error("zkhip_create_proof: cs_degree %u < 3", pk->cs_degree); return ZKHIP_EINVAL; }
    const uint32_t chunk = pk->cs_degree - 2;
    const uint32_t Zp = P ? (P + chunk - 1) / chunk : 0;
    if (zkhip_domain_k(pk->domain) != k || zkhip_srs_len(pk->g) < n || zkhip_srs_len(pk->g_lagrange) < n) {
        set_error("zkhip_create_proof: SRS / domain do not match k = %u", k);
        return ZKHIP_EINVAL;
    }
    // every input check comes BEFORE the first asynchronous copy from the caller's memory
    if (!in->d_instance)
        for (uint32_t j = 0; j < I; ++j)
            if (in->instance_len[j] > n || (in->instance_len[j] && !in->instance_values[j])) {
                set_error("zkhip_create_proof: instance column %u has %zu values (n = %zu) or a null pointer", j, (size_t)in->instance_len[j], n);
                return ZKHIP_EINVAL;
            }
    // advice phases / user challenges (zk_proving_key.advice_column_phase, challenge_phase): phase 0 only unless the key says otherwise
    const uint32_t NC = pk->n_challenges;
    auto phase_of = [&](uint32_t j) -> uint32_t { return pk->advice_column_phase ? pk->advice_column_phase[j] : 0u; };
    uint32_t max_phase = 0;
    for (uint32_t j = 0; j < A; ++j) max_phase = std::max(max_phase, phase_of(j));
    if (NC && !pk->challenge_phase) { set_error("zkhip_create_proof: n_challenges = %u but challenge_phase is NULL", NC); return ZKHIP_EINVAL; }
    for (uint32_t c = 0; c < NC; ++c) max_phase = std::max<uint32_t>(max_phase, pk->challenge_phase[c]);
    {
        // the phases of a circuit are the phases of its advice columns, without gaps (upstream's ConstraintSystem hands them out in order and a
        // challenge is usable "after" a phase that exists); the callback is needed for every phase after the first
        uint32_t max_adv = 0;
        for (uint32_t j = 0; j < A; ++j) max_adv = std::max(max_adv, phase_of(j));
        if (max_adv >= 64) { set_error("zkhip_create_proof: advice phase %u (at most 63)", max_adv); return ZKHIP_EINVAL; }
        uint64_t seen = 0;
        for (uint32_t j = 0; j < A; ++j) seen |= 1ull << phase_of(j);
        if (A && seen != ((max_adv == 63) ? ~0ull : ((1ull << (max_adv + 1)) - 1))) {
            set_error("zkhip_create_proof: advice_column_phase has a phase without columns (phases 0..%u must all occur)", max_adv);
            return ZKHIP_EINVAL;
        }
        for (uint32_t c = 0; c < NC; ++c)
            if (pk->challenge_phase[c] > max_adv) {
                set_error("zkhip_create_proof: challenge %u is of phase %u but the last advice phase is %u", c, (unsigned)pk->challenge_phase[c], max_adv);
                return ZKHIP_EINVAL;
            }
        if (max_phase > 0 && !in->advice_phase) {
            set_error("zkhip_create_proof: the key has phases up to %u but zk_proof_inputs.advice_phase is NULL", max_phase);
            return ZKHIP_EINVAL;
        }
    }
    const bool multi_phase = max_phase > 0 || NC > 0;
    for (uint32_t j = 0; j < A; ++j)
        if (phase_of(j) == 0 && !in->advice[j]) { set_error("zkhip_create_proof: advice column %u is null", j); return ZKHIP_EINVAL; }
    std::vector<uint64_t> user_ch(4 * (size_t)std::max<uint32_t>(NC, 1), 0);   // the user challenges, ABI form, index = challenge index
    if (bl && ((L && !bl->lookup_permuted) || (Zp && bf && !bl->perm_z) || (L && bf && !bl->lookup_z) || !bl->random_poly)) {
        // a caller that supplies its rng draws supplies ALL of them: silently falling back to the seeded generator for a missing member
        // would make those blinding rows predictable (zero knowledge lost without an error)
        set_error("zkhip_create_proof: zk_blinding given but a member with a non-zero row count is NULL (lookup_permuted / perm_z / lookup_z / random_poly)");
        return ZKHIP_EINVAL;
    }
    uint64_t omega_abi[4], ext_omega_abi[4], g_coset_abi[4];
    zkhip_domain_constants(pk->domain, omega_abi, ext_omega_abi, g_coset_abi);

    // second stream for the overlapped coset NTTs (owned by the context, created on first use)
    if (!ctx->side_stream) {
        ZK_HIP(hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
        ZK_HIP(hipEventCreateWithFlags(&ctx->side_event, hipEventDisableTiming));
    }
    const bool late = ctx->opt.late_overlap >= 0 ? ctx->opt.late_overlap != 0 : k <= 19;   // re-measured with the 4-per-thread NTT kernels: k = 19 -1.4 %, k = 22 +0.8 %
    const bool serial = ctx->opt.late_overlap == 2;   // analysis only: everything on the main stream, so a kernel trace shows isolated durations
    Overlap ov{ctx, ctx->stream, serial ? ctx->stream : ctx->side_stream, ctx->side_event, late};
    StreamGuard guard{ctx, ctx->stream};
    guard.host_uploads = in->advice_on_host || (bl && bl->on_host);
    hipStream_t st = ctx->stream;

    // ---- one proof over several GPUs (comm.hip): MSMs are collective by themselves (sharded SRS handles); here the coset NTTs go by
    // polynomial and the quotient sweep by row range, both all-gathered in place.  Everything else is replicated: every rank runs this
    // same schedule on identical inputs and ends with the identical transcript.
    const size_t NR = (size_t)ctx->comm.nranks, RK = (size_t)ctx->comm.rank;
    const bool dist = NR > 1;
    auto pad = [&](size_t cnt) { return dist ? (cnt + NR - 1) / NR * NR : cnt; };   // all-gather rounds of NR columns: pad the column blocks

    // ---- workspace (library-owned, reused across proofs)
    // ---- the quotient's evaluation points: the extended domain, or — when that is fewer rows and the key is known to the context
    // (key_id) — quotient_poly_degree cosets of the size-n domain (cosets.hip); the same h either way
    const zk::CosetPlan* cplan = nullptr;
    const zk::KeyCosets* kcos = nullptr;
    const bool coset_mode = zkhip_coset_quotient_applies(ctx, pk) != 0;
    if (coset_mode) {
        ZK_TRY(zk::coset_plan(ctx, pk->domain, &cplan));
        ZK_TRY(zk::key_cosets(ctx, cplan, pk, &kcos));
    } else if ((F && !pk->fixed_cosets) || (P && !pk->sigma_cosets) || ((P || L) && (!pk->l0 || !pk->l_last || !pk->l_active_row))) {
        set_error("zkhip_create_proof: the proving key's extended cosets are missing (they are optional only with key_id and coefficient forms)");
        return ZKHIP_EINVAL;
    }
    const size_t ext_rows = coset_mode ? (size_t)qd * n : en;
    char *w_coeff, *w_ext, *w_rand, *w_comp, *w_blind, *w_perm_l, *w_perm_c, *w_ext_perm, *w_z, *w_ext_z, *w_h, *w_hvals = nullptr, *w_hpoly, *w_evals, *w_com;
    const size_t NB = n * 32, EB = ext_rows * 32;
    auto ws = [&](const char* name, size_t bytes, char** p) { void* q; int rc = ctx->get_scratch(name, bytes ? bytes : 32, &q); *p = (char*)q; return rc; };
    ZK_TRY(ws("cp_coeff", (A + I) * NB, &w_coeff));
    ZK_TRY(ws("cp_ext", pad(A + I) * EB, &w_ext));
    ZK_TRY(ws("cp_rand", NB, &w_rand));
    // host-side inputs (a Rust caller's Vec<Fr> columns, its instance values, its rng draws) are uploaded into library-owned columns
    char *w_adv_in = nullptr, *w_ins_in = nullptr;
    if (in->advice_on_host && A) ZK_TRY(ws("cp_adv_in", (size_t)A * NB, &w_adv_in));
    if (!in->d_instance && I) ZK_TRY(ws("cp_ins_in", (size_t)I * NB, &w_ins_in));
    std::vector<const void*> adv_cols(A), ins_cols(I);
    // Large host columns (>= 64 MiB in all: 512 MiB at k = 22 = 10 ms over PCIe) are uploaded on a copy stream of their own while the
    // main stream already commits the vanishing argument's random polynomial, which needs none of them (phase 1 below); small ones
    // ride on the proof's stream.  From pinned memory the copies run at link rate; from pageable memory the runtime stages them.
    const bool split_upload = !multi_phase && in->advice_on_host && A && (size_t)A * NB >= ((size_t)64 << 20);
    // The vanishing argument's random polynomial depends on nothing: its commitment can leave the advice batch and run on a third stream
    // beside the grand products and the z columns' inverse transforms (memory-bound kernels on the main stream).  Measured A/B (round 4,
    // gpurun_out/r04p): k = 17 7.07 / 7.09 -> 6.94 / 6.99 ms (the MSM's latency chain hides behind the products phase), k = 19 neutral,
    // k = 22 +0.25 ms (the side stream's NTTs already fill every idle multiplier slot: the proof is the sum of its instruction streams) —
    // so by default only where proofs are latency-shaped (k <= 18).  Needs a products phase (Zp + L > 0).
    const bool rand_late = !multi_phase && !split_upload && (Zp + L) > 0 && (ctx->opt.rand_overlap < 0 ? k <= 18 : ctx->opt.rand_overlap != 0);
    if (rand_late && !ctx->aux_stream) {
        ZK_HIP(hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking));
        for (auto& e : ctx->aux_event) ZK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    // many columns (the SHA-256 circuit's 32): uploaded and committed in up to 4 groups of >= 8 columns, so that the commitment of group
    // g runs while group g + 1 is still on the wire; few big columns (the aggregation circuit's 4): one group behind the random polynomial
    // few columns of >= 64 MiB each (k >= 21: a column's upload, 2.5 ms, is shorter than its commitment): groups of two columns, up to 4
    // groups — every extra batch costs a host round trip and a latency-bound tail, so no finer
    const uint32_t n_groups = !split_upload ? 1 : NB >= ((size_t)64 << 20) ? std::min<uint32_t>(4, (A + 1) / 2) : std::max<uint32_t>(1, std::min<uint32_t>(4, A / 8));
    auto group_begin = [&](uint32_t g_) { return (uint32_t)((uint64_t)A * g_ / n_groups); };
    bool rand_on_copy_stream = false;
    size_t rand_chunks = 1, rand_first_job = 0;
    // The large uploads of the caller's host buffers: a list of copy jobs on the copy stream — the random polynomial first (it is what the main stream needs first), then the
    // advice columns group by group, each group's last copy followed by the group's event.  With option host_copy_thread (default) the list is executed by a worker thread
    // (zk_copy_worker, common.hpp): pageable sources — a Rust caller's Vec<Fr> — block THAT thread for their copies' duration while this one goes on launching; measured with
    // FRESH pageable columns per proof: SHA-shaped k = 19 45.2 ms from this thread, 58-60 ms with the buffers registered first (cold hipHostRegister: ~0.7 ms per call), and the
    // pinned figure with the worker (profiles/r06_host_inputs.txt).  Before the main stream is made to wait for a job's event, this thread waits for the worker to have recorded it.
    zk_copy_worker upload_worker;
    std::vector<size_t> group_job(n_groups, 0);
    const bool threaded = split_upload && ctx->opt.host_copy_thread != 0;
    if (split_upload) {
        if (!ctx->copy_stream) {
            ZK_HIP(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
            for (auto& e : ctx->copy_event) ZK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        ZK_HIP(hipEventRecord(ctx->copy_event[0], st));   // the upload buffer's last readers (the previous proof) were issued on the main stream
        ZK_HIP(hipStreamWaitEvent(ctx->copy_stream, ctx->copy_event[0], 0));
        // The caller's random polynomial (zk_blinding on the host: n scalars drawn from ITS rng — 128 MiB at k = 22) is the FIRST thing the main stream needs: it goes over the
        // link first, on the copy stream — as a pageable hipMemcpyAsync on the main stream (until round 6) it blocked the host for 2.4 ms at the top of the proof while
        // competing with the advice uploads for the link (k = 22 with host advice + host blinding: +5.8 ms) — in K chunks when it is large: its commitment, the proof's first
        // MSM, then runs chunk by chunk behind the bytes (msm.hip host_column_*: the pipeline of zkhip_msm_g1)
        if (bl && bl->on_host && bl->random_poly && NB >= ((size_t)16 << 20)) {
            if (ctx->opt.host_register != 0) guard.pin(bl->random_poly, NB);
            rand_chunks = zk::host_column_chunks(ctx, pk->g, n);
            rand_first_job = upload_worker.jobs.size();
            if (rand_chunks > 1) {
                ZK_TRY(zk::host_column_jobs(ctx, bl->random_poly, n, rand_chunks, w_rand, &upload_worker.jobs));
            } else {
                if (!ctx->copy_event_rand) ZK_HIP(hipEventCreateWithFlags(&ctx->copy_event_rand, hipEventDisableTiming));
                upload_worker.jobs.push_back(zk_copy_job{w_rand, bl->random_poly, NB, ctx->copy_event_rand});
            }
            rand_on_copy_stream = true;
        }
    } else if (bl && bl->on_host && bl->random_poly && NB >= ((size_t)16 << 20) && ctx->opt.host_register != 0) {
        guard.pin(bl->random_poly, NB);      // (uploaded on the main stream below: registered, the copy does not hold the host)
    }
    for (uint32_t g_ = 0; g_ < n_groups; ++g_) {
        for (uint32_t j = group_begin(g_); j < group_begin(g_ + 1); ++j) {
            if (phase_of(j) != 0) continue;     // bound after its phase's witness exists (bind_phase below)
            if (in->advice_on_host) {
                if (split_upload) {
                    if (ctx->opt.host_register != 0) guard.pin(in->advice[j], NB);
                    upload_worker.jobs.push_back(zk_copy_job{w_adv_in + j * NB, in->advice[j], NB, nullptr});
                } else {
                    ZK_HIP(hipMemcpyAsync(w_adv_in + j * NB, in->advice[j], NB, hipMemcpyHostToDevice, st));
                }
                adv_cols[j] = w_adv_in + j * NB;
            } else {
                adv_cols[j] = in->advice[j];
            }
        }
        if (split_upload) {      // the group's event behind its last copy (an empty job if the group had none)
            upload_worker.jobs.push_back(zk_copy_job{nullptr, nullptr, 0, ctx->copy_event[g_]});
            group_job[g_] = upload_worker.jobs.size() - 1;
        }
    }
    if (split_upload) {
        // (the worker is declared after `guard`: on every exit it is joined — its destructor — BEFORE the guard drains the copy stream and gives the buffers back)
        if (!threaded || !upload_worker.start(ctx->device, ctx->copy_stream))
            ZK_HIP(upload_worker.run_inline(ctx->copy_stream));      // from this thread, in order (pageable sources block it copy by copy); wait() then returns at once
    }
    auto upload_issued = [&](size_t job) -> int {      // host side: the worker has issued job `job` (and recorded its event)
        if (split_upload) ZK_HIP(upload_worker.wait(job));
        return ZKHIP_OK;
    };
    for (uint32_t j = 0; j < I; ++j) {
        if (in->d_instance) { ins_cols[j] = in->d_instance[j]; continue; }
        const size_t len = in->instance_len[j];
        ZK_HIP(hipMemsetAsync(w_ins_in + j * NB, 0, NB, st));
        if (len) ZK_TRY(ctx->upload(w_ins_in + j * NB, in->instance_values[j], len * 32));
        ins_cols[j] = w_ins_in + j * NB;
    }
    const void* const* d_advice = adv_cols.data();
    const void* const* d_instance = ins_cols.data();
    // the caller's rng draws (or the seeded stand-ins)
    auto blind_rows = [&](const void* src, size_t offset_elems, size_t count, char* dst, uint64_t seed) -> int {
        if (!count) return ZKHIP_OK;
        if (bl && src) {
            const char* from = (const char*)src + offset_elems * 32;
            if (bl->on_host) return ctx->upload(dst, from, count * 32);
            ZK_HIP(hipMemcpyAsync(dst, from, count * 32, hipMemcpyDeviceToDevice, ctx->stream));
            return ZKHIP_OK;
        }
        return zkhip_synth_fill_device(ctx, dst, count, seed, 0);
    };
    ZK_TRY(ws("cp_comp", 2 * L * NB, &w_comp));
    ZK_TRY(ws("cp_blind", (2 * L * (bf + 1) + (Zp + L) * bf + 8) * 32, &w_blind));
    ZK_TRY(ws("cp_perm_l", 2 * L * NB, &w_perm_l));
    ZK_TRY(ws("cp_perm_c", 2 * L * NB, &w_perm_c));
    ZK_TRY(ws("cp_ext_perm", pad(2 * L) * EB, &w_ext_perm));
    ZK_TRY(ws("cp_z", (Zp + L) * NB, &w_z));
    ZK_TRY(ws("cp_ext_z", pad(Zp + L) * EB, &w_ext_z));
    ZK_TRY(ws("cp_h", EB, &w_h));
    if (coset_mode) ZK_TRY(ws("cp_hvals", EB, &w_hvals));
    ZK_TRY(ws("cp_hpoly", NB, &w_hpoly));
    const size_t max_q = (size_t)pk->n_advice_queries + pk->n_fixed_queries + 3 * Zp + 5 * L + P + 2;
    ZK_TRY(ws("cp_evals", max_q * 32, &w_evals));
    ZK_TRY(ws("cp_com", (A + 2 * L + Zp + L + qd + 2) * 96, &w_com));
    // commitments are latency-critical read-backs: let the MSM's last kernel store them straight into pinned host memory
    // (4 KiB in, past the small read-back slots) when they fit
    if ((A + 2 * L + Zp + L + qd + 2) * 96 + 4096 + 64 <= zkhip_ctx::PINNED_BYTES) w_com = (char*)ctx->h_pinned + 4096;

    // ---- row-sharded cosets: with the quotient on q cosets of the size-n domain, rank R sweeps rows [R m, (R+1) m),
    // m = n / N, of EVERY coset block, and a rotation stays inside its block — so of every column it needs those rows plus a halo of
    // H = max |rotation| rows on either side (modulo n), and nothing else.  The coset NTTs stay by polynomial (the rank that transforms a
    // column holds all of it); instead of all-gathering complete columns, each owner sends every peer ITS row windows: an all-to-all of
    // q (m + 2 H) rows per (column, peer) = 1 / N of the all-gather's volume.  The column buffers keep their full q n size (the sweep
    // addresses them as before; the memory is there), a rank just never fills — or reads — the rows outside its windows.
    uint32_t halo = std::max<uint32_t>(1, bf + 1);
    auto scan_rots = [&](const zk_graph& g_) { for (uint32_t i = 0; i < g_.n_rotations; ++i) halo = std::max<uint32_t>(halo, (uint32_t)std::abs(g_.rotations[i])); };
    scan_rots(pk->custom_gates);
    for (uint32_t i = 0; i < L; ++i) scan_rots(pk->lookup_graphs[i]);
    const size_t m_rows = n / NR;
    const bool row_mode = dist && coset_mode && ctx->comm.row_sharded(ctx->opt) && n % (NR * 64) == 0 && 2 * (size_t)halo < m_rows;
    // With MSMs by point range a commitment reads only this rank's rows of a column: the quotient's pieces, h(X) and SHPLONK's polynomials
    // then never need to be complete anywhere (shplonk.hip makes the same test on the SRS handle).
    bool pieces_sharded = false;
    if (row_mode && !ctx->comm.shard_columns) {
        size_t s_first, s_count, s_total;
        zkhip_srs_range(pk->g, &s_first, &s_count, &s_total);
        pieces_sharded = s_total == n && s_count == m_rows && s_first == RK * m_rows && m_rows >= 64;
    }
    if (row_mode) ctx->n_row_sharded += 1;
    if (pieces_sharded) ctx->n_pieces_sharded += 1;
    const size_t my_lo = pieces_sharded ? RK * m_rows : 0, my_n = pieces_sharded ? m_rows : n, my_lo_b = my_lo * 32;
    // Owners (row-sharded exchange only).  Column j of a batch is transformed by ONE rank — the owner — and the batches of a proof are small
    // (A + I, 2 L, Zp + L columns, q numerator blocks: 5, 2, 5, 3 for the aggregation shape): with owner = j mod N rank 0 owned a column of
    // every batch and ranks 5-7 of 8 none (single-rank replay, round 5: 9.7 ms of transforms on rank 0, 5.2 on rank 3, none on rank 7).
    // The batches are therefore dealt round robin ACROSS batches: owner(j) = (j + rot) mod N with rot = the columns dealt so far, so no
    // rank owns more than ceil(total / N) columns of a proof (15 over 8 ranks: two each, one rank one).  The coset transforms of a batch
    // float past the phase that issued them (the sweep is their first reader), so what a rank's GPU spends on them adds to ITS proof time
    // wherever it falls — spreading them is what shortens the slowest rank.  first_of(rot, r) = the first column of rank r in a batch.
    const size_t rot_adv = 0, rot_perm = dist ? (A + I) % NR : 0, rot_z = dist ? (A + I + 2 * L) % NR : 0, rot_h = dist ? (A + I + 2 * L + Zp + L) % NR : 0;
    auto first_of = [&](size_t rot, size_t r) -> size_t { return (r + NR - rot % NR) % NR; };
    auto owner_of = [&](size_t rot, size_t j) -> size_t { return (j + rot) % NR; };
    // coeff_to_extended of `count` polynomials whose outputs are consecutive EB-sized slices of one padded workspace block
    auto to_extended = [&](const void* const* srcs, void* const* dsts, size_t count, size_t rot) -> int {
        auto transform = [&](const void* const* s_, void* const* d_, size_t c_) -> int {
            return coset_mode ? zk::coeff_to_cosets(ctx, cplan, s_, d_, c_) : zkhip_coeff_to_extended_device(ctx, pk->domain, s_, n, d_, c_);
        };
        if (!dist) return transform(srcs, dsts, count);
        if (row_mode) {
            // my columns (j = first_of(rot, RK) + t NR) in one batch, then the windows of every peer packed, exchanged and unpacked
            std::vector<const void*> ms;
            std::vector<void*> md;
            for (size_t j = first_of(rot, RK); j < count; j += NR) { ms.push_back(srcs[j]); md.push_back(dsts[j]); }
            if (!ms.empty()) ZK_TRY(transform(ms.data(), md.data(), ms.size()));
            const size_t maxcols = (count + NR - 1) / NR, W = m_rows + 2 * halo, blk = maxcols * qd * W * 32;
            char *w_send, *w_recv;
            ZK_TRY(ws("cp_a2a_send", NR * blk, &w_send));
            ZK_TRY(ws("cp_a2a_recv", NR * blk, &w_recv));
            const uint32_t nmask = (uint32_t)n - 1;
            std::vector<zk::RowCopy> list;
            for (size_t r = 0; r < NR; ++r) {
                if (r == RK) continue;
                for (size_t t = 0; t < md.size(); ++t)
                    for (uint32_t b_ = 0; b_ < qd; ++b_)
                        list.push_back(zk::RowCopy{(const uint32_t*)((char*)md[t] + (size_t)b_ * NB), (uint32_t*)(w_send + r * blk + (t * qd + b_) * W * 32),
                                                   (uint32_t)((r * m_rows + n - halo) & nmask), 0u, (uint32_t)W, nmask, 0xffffffffu});
            }
            ZK_TRY(zk::comm_row_copies(ctx, list));
            std::vector<uint8_t> has_cols(NR), to_all(NR, md.empty() ? 0 : 1);
            for (size_t r = 0; r < NR; ++r) has_cols[r] = first_of(rot, r) < count;      // rank r transforms columns first_of(rot, r), + NR, ...: maybe none
            // (on the bulk communicator when the context has one: the sweep is these windows' first reader, nothing here is latency-critical)
            ZK_TRY(zk::comm_alltoall(ctx, w_send, w_recv, blk, to_all.data(), has_cols.data(), true));
            list.clear();
            for (size_t r = 0; r < NR; ++r) {
                if (r == RK) continue;
                size_t t = 0;
                for (size_t j = first_of(rot, r); j < count; j += NR, ++t)
                    for (uint32_t b_ = 0; b_ < qd; ++b_)
                        list.push_back(zk::RowCopy{(const uint32_t*)(w_recv + r * blk + (t * qd + b_) * W * 32), (uint32_t*)((char*)dsts[j] + (size_t)b_ * NB), 0u,
                                                   (uint32_t)((RK * m_rows + n - halo) & nmask), (uint32_t)W, 0xffffffffu, nmask});
            }
            return zk::comm_row_copies(ctx, list);
        }
        // by polynomial, pipelined: rank RK transforms column t NR + RK of round t, then the round's NR columns are all-gathered in
        // place on the communicator's stream while this stream already transforms the rank's column of round t + 1
        for (size_t t = 0; t * NR < count; ++t) {
            const size_t j = t * NR + RK;
            if (j < count) {
                const void* s1[1] = {srcs[j]};
                void* d1[1] = {dsts[j]};
                ZK_TRY(transform(s1, d1, 1));
            }
            char* block = (char*)dsts[t * NR];
            ZK_TRY(zk::comm_allgather_begin(ctx, block + RK * EB, block, EB));
        }
        ZK_TRY(zk::comm_allgather_end(ctx));
        return ZKHIP_OK;
    };

    // ---- owner / row-range layout of the NEW columns (pieces_sharded only).  Column j of a batch has the owner j mod N — the rank that runs
    // its inverse transform, its coset NTT and its evaluations and therefore holds it COMPLETE — and every rank holds ITS ROW RANGE of the
    // coefficient form, which is all a point-range commitment and SHPLONK on row ranges read.  exchange_ranges moves row ranges between
    // the ranks and the owners: dir 0 = every rank's rows of column j to owner(j) (who ends up with the whole column), dir 1 = the owner's
    // rows [r m, (r + 1) m) of column j to rank r.
    auto exchange_ranges = [&](int dir, const void* const* src, void* const* dst, size_t count, size_t rot, bool bulk = false) -> int {
        const size_t maxcols = (count + NR - 1) / NR, blk = maxcols * m_rows * 32;
        char *w_send, *w_recv;
        ZK_TRY(ws("cp_a2a_send", NR * blk, &w_send));
        ZK_TRY(ws("cp_a2a_recv", NR * blk, &w_recv));
        const uint32_t full = 0xffffffffu, M = (uint32_t)m_rows;
        std::vector<uint8_t> is_owner(NR), all1(NR, 1), all0(NR, 0);
        for (size_t r = 0; r < NR; ++r) is_owner[r] = first_of(rot, r) < count;
        const bool i_own = first_of(rot, RK) < count;
        std::vector<zk::RowCopy> list;
        auto rows = [&](const void* base, size_t row) { return (const uint32_t*)((const char*)base + row * 32); };
        for (size_t r = 0; r < NR; ++r) {
            if (r == RK) continue;
            if (dir == 0) { size_t t = 0; for (size_t j = first_of(rot, r); j < count; j += NR, ++t) list.push_back(zk::RowCopy{rows(src[j], RK * m_rows), (uint32_t*)(w_send + r * blk + t * m_rows * 32), 0u, 0u, M, full, full}); }
            else { size_t t = 0; for (size_t j = first_of(rot, RK); j < count; j += NR, ++t) list.push_back(zk::RowCopy{rows(src[j], r * m_rows), (uint32_t*)(w_send + r * blk + t * m_rows * 32), 0u, 0u, M, full, full}); }
        }
        ZK_TRY(zk::comm_row_copies(ctx, list));
        if (dir == 0) ZK_TRY(zk::comm_alltoall(ctx, w_send, w_recv, blk, is_owner.data(), i_own ? all1.data() : all0.data(), bulk));
        else ZK_TRY(zk::comm_alltoall(ctx, w_send, w_recv, blk, i_own ? all1.data() : all0.data(), is_owner.data(), bulk));
        list.clear();
        for (size_t r = 0; r < NR; ++r) {
            if (r == RK) continue;
            if (dir == 0) { size_t t = 0; for (size_t j = first_of(rot, RK); j < count; j += NR, ++t) list.push_back(zk::RowCopy{(const uint32_t*)(w_recv + r * blk + t * m_rows * 32), (uint32_t*)rows(dst[j], r * m_rows), 0u, 0u, M, full, full}); }
            else { size_t t = 0; for (size_t j = first_of(rot, r); j < count; j += NR, ++t) list.push_back(zk::RowCopy{(const uint32_t*)(w_recv + r * blk + t * m_rows * 32), (uint32_t*)rows(dst[j], RK * m_rows), 0u, 0u, M, full, full}); }
        }
        for (size_t j = first_of(rot, RK); j < count; j += NR)      // the owner's own rows when the exchange is not in place
            if (src[j] != dst[j]) list.push_back(zk::RowCopy{rows(src[j], RK * m_rows), (uint32_t*)rows(dst[j], RK * m_rows), 0u, 0u, M, full, full});
        return zk::comm_row_copies(ctx, list);
    };
    // lagrange_to_coeff of a batch by owner: lag[j] complete on every rank (lag_sharded = false) or present as row ranges only (true);
    // afterwards coeff[j] is complete on owner(j) and every rank holds its row range of it
    // bulk: nothing latency-critical reads the result (the advice batch: its commitments are over the Lagrange basis; the coefficient ranges
    // are for the evaluations and SHPLONK, much later) — the ranges travel on the bulk communicator beside the row windows
    auto owner_intt = [&](const void* const* lag, void* const* coeff, size_t count, bool lag_sharded, size_t rot, bool bulk = false) -> int {
        if (lag_sharded) ZK_TRY(exchange_ranges(0, lag, coeff, count, rot, bulk));
        std::vector<const void*> ms;
        std::vector<void*> md;
        for (size_t j = first_of(rot, RK); j < count; j += NR) { ms.push_back(lag_sharded ? (const void*)coeff[j] : lag[j]); md.push_back(coeff[j]); }
        if (!ms.empty()) ZK_TRY(zk::lagrange_to_coeff_oop(ctx, pk->domain, ms.data(), md.data(), ms.size()));
        return exchange_ranges(1, (const void* const*)coeff, coeff, count, rot, bulk);
    };

    uint64_t ch[4];
    std::vector<uint64_t> xy;
    std::vector<uint8_t> by;
    // commit a batch: MSMs, read back, hand every point to the transcript (skip_last: committed now, written later)
    auto commit_launch = [&](const std::vector<const void*>& cols, const std::vector<const zkhip_srs*>& bases) -> int {
        if (cols.empty()) return ZKHIP_OK;
        return zkhip_msm_g1_multi_device(ctx, bases.data(), cols.data(), cols.size(), 0, n, w_com);
    };
    auto commit_read = [&](size_t m, size_t hold_back, std::vector<uint64_t>* held_xy, std::vector<uint8_t>* held_by) -> int {
        if (!m) return ZKHIP_OK;
        xy.resize(8 * m);
        by.resize(32 * m);
        ZK_TRY(zkhip_commitments_read(ctx, w_com, m, xy.data(), by.data()));
        for (size_t j = 0; j + hold_back < m; ++j) tr->write_point(tr->user, by.data() + 32 * j, xy.data() + 8 * j);
        if (hold_back && held_xy) { held_xy->assign(xy.begin() + 8 * (m - hold_back), xy.end()); held_by->assign(by.begin() + 32 * (m - hold_back), by.end()); }
        return ZKHIP_OK;
    };
    auto commit = [&](const std::vector<const void*>& cols, const std::vector<const zkhip_srs*>& bases, size_t hold_back,
                      std::vector<uint64_t>* held_xy, std::vector<uint8_t>* held_by) -> int {
        ZK_TRY(commit_launch(cols, bases));
        return commit_read(cols.size(), hold_back, held_xy, held_by);
    };

    // ---- 1. advice (+ the vanishing argument's random polynomial, which depends on no challenge) ; coset NTTs of advice/instance overlap
    if (rand_on_copy_stream) {
        if (rand_chunks <= 1) {      // (in chunks: the commitment below waits chunk by chunk)
            ZK_TRY(upload_issued(rand_first_job));
            ZK_HIP(hipStreamWaitEvent(st, ctx->copy_event_rand, 0));
        }
    } else if (bl && bl->random_poly) {
        if (bl->on_host) ZK_HIP(hipMemcpyAsync(w_rand, bl->random_poly, NB, hipMemcpyHostToDevice, st));
        else ZK_HIP(hipMemcpyAsync(w_rand, bl->random_poly, NB, hipMemcpyDeviceToDevice, st));
    } else {
        ZK_TRY(zkhip_synth_fill_device(ctx, w_rand, n, blinding_seed + 380, 0));
    }

    if (rand_late) ZK_HIP(hipEventRecord(ctx->aux_event[0], st));   // w_rand is complete on the main stream from here on
    std::vector<void*> coeff_ptrs(A + I), ext_ptrs(A + I);
    for (uint32_t j = 0; j < A + I; ++j) { coeff_ptrs[j] = w_coeff + j * NB; ext_ptrs[j] = w_ext + j * EB; }
    std::vector<uint64_t> rand_xy;
    std::vector<uint8_t> rand_by;
    auto absorb_vk_and_instances = [&]() {
        // vk.hash_into(transcript), then every instance value (KZG: hashed, not committed) — upstream's first transcript operations.
        // Nothing has entered the transcript yet, so they are absorbed while the GPU works on the first advice commitments (a sponge
        // transcript such as Poseidon spends ~10 us per absorbed pair: 32 instance values would otherwise sit on the critical path).
        if (!tr->common_scalar) return;
        if (pk->vk_transcript_repr) tr->common_scalar(tr->user, pk->vk_transcript_repr);
        if (in->instance_values)
            for (uint32_t j = 0; j < I; ++j)
                for (uint32_t i = 0; i < in->instance_len[j]; ++i) tr->common_scalar(tr->user, in->instance_values[j] + 4 * i);
    };
    if (multi_phase) {
        // Upstream's loop over the phases (plonk/prover.rs [UPSTREAM-RECALL]): the witness of phase p (the caller's, through advice_phase:
        // it depends on the challenges of the earlier phases), the commitments of phase p's columns — in column order, the random
        // polynomial riding with phase 0 — written to the transcript, then the user challenges whose phase is p.  One commitment batch
        // and one Fiat-Shamir round trip per phase; the columns' transforms follow the last phase (they overlap the lookup commitments).
        const void** adv_in = const_cast<const void**>(in->advice);
        for (uint32_t ph = 0; ph <= max_phase; ++ph) {
            if (ph > 0) {
                if (in->advice_phase(in->advice_phase_user, ph, user_ch.data(), adv_in) != 0) {
                    set_error("zkhip_create_proof: the caller's advice_phase callback failed for phase %u", ph);
                    return ZKHIP_EINVAL;
                }
                for (uint32_t j = 0; j < A; ++j) {
                    if (phase_of(j) != ph) continue;
                    if (!adv_in[j]) { set_error("zkhip_create_proof: advice column %u (phase %u) is null after advice_phase", j, ph); return ZKHIP_EINVAL; }
                    if (in->advice_on_host) {
                        ZK_HIP(hipMemcpyAsync(w_adv_in + j * NB, adv_in[j], NB, hipMemcpyHostToDevice, st));
                        adv_cols[j] = w_adv_in + j * NB;
                    } else {
                        adv_cols[j] = adv_in[j];
                    }
                }
            }
            std::vector<const void*> cols;
            std::vector<const zkhip_srs*> bases;
            for (uint32_t j = 0; j < A; ++j)
                if (phase_of(j) == ph) { cols.push_back(adv_cols[j]); bases.push_back(pk->g_lagrange); }
            if (ph == 0) { cols.push_back(w_rand); bases.push_back(pk->g); }
            ZK_TRY(commit_launch(cols, bases));
            if (ph == 0) absorb_vk_and_instances();
            ZK_TRY(commit_read(cols.size(), ph == 0 ? 1 : 0, &rand_xy, &rand_by));
            for (uint32_t c = 0; c < NC; ++c)
                if (pk->challenge_phase[c] == ph) tr->squeeze_challenge(tr->user, user_ch.data() + 4 * (size_t)c);
        }
        ZK_TRY(ov.begin());
        if (A + I) {
            std::vector<const void*> lag(A + I);
            for (uint32_t j = 0; j < A + I; ++j) lag[j] = j < A ? d_advice[j] : d_instance[j - A];
            if (pieces_sharded) ZK_TRY(owner_intt(lag.data(), coeff_ptrs.data(), A + I, false, rot_adv, true));
            else ZK_TRY(zk::lagrange_to_coeff_oop(ctx, pk->domain, lag.data(), coeff_ptrs.data(), A + I));
            ZK_TRY(to_extended((const void* const*)coeff_ptrs.data(), ext_ptrs.data(), A + I, rot_adv));
        }
        ov.end();
    } else {
        std::vector<const void*> cols(d_advice, d_advice + A);
        std::vector<const zkhip_srs*> bases(A, pk->g_lagrange);
        if (!rand_late) {
            cols.push_back(w_rand);
            bases.push_back(pk->g);
        }
        if (split_upload) {
            // the random polynomial's commitment first (its slot is the last one of the batch), then — once the uploads have landed —
            // the advice columns'
            const void* rc[1] = {w_rand};
            const zkhip_srs* rb[1] = {pk->g};
            if (rand_chunks > 1) ZK_TRY(zk::host_column_commit(ctx, pk->g, n, rand_chunks, w_rand, w_com + (size_t)A * 96, &upload_worker, rand_first_job));
            else ZK_TRY(zkhip_msm_g1_multi_device(ctx, rb, rc, 1, 0, n, w_com + (size_t)A * 96));
            for (uint32_t g_ = 0; g_ < n_groups; ++g_) {
                const uint32_t j0 = group_begin(g_), j1 = group_begin(g_ + 1);
                ZK_TRY(upload_issued(group_job[g_]));
                ZK_HIP(hipStreamWaitEvent(st, ctx->copy_event[g_], 0));
                if (g_ + 1 == n_groups) ov.arm();   // the overlapped NTTs follow the last group's accumulation
                ZK_TRY(zkhip_msm_g1_multi_device(ctx, bases.data() + j0, cols.data() + j0, j1 - j0, 0, n, w_com + (size_t)j0 * 96));
            }
        } else {
            ov.arm();
            ZK_TRY(commit_launch(cols, bases));
        }
        ZK_TRY(ov.begin_marked());
        if (A + I) {
            std::vector<const void*> lag(A + I);   // out of place: the witness columns stay in Lagrange form, no copy
            for (uint32_t j = 0; j < A + I; ++j) lag[j] = j < A ? d_advice[j] : d_instance[j - A];
            if (pieces_sharded) ZK_TRY(owner_intt(lag.data(), coeff_ptrs.data(), A + I, false, rot_adv, true));
            else ZK_TRY(zk::lagrange_to_coeff_oop(ctx, pk->domain, lag.data(), coeff_ptrs.data(), A + I));
            ZK_TRY(to_extended((const void* const*)coeff_ptrs.data(), ext_ptrs.data(), A + I, rot_adv));
        }
        ov.end();
        absorb_vk_and_instances();
        mark("vk + instances absorbed (GPU busy)");
        ZK_TRY(commit_read(cols.size(), rand_late ? 0 : 1, &rand_xy, &rand_by));
    }
    mark("advice committed + absorbed");
    tr->squeeze_challenge(tr->user, ch);
    mark("theta");
    ctx->comm.phase = "lookup permute";
    uint64_t theta[4];
    memcpy(theta, ch, 32);

    // ---- 2. lookups: compression (the sweep interpreter on the Lagrange domain), permuted columns, their commitments
    std::vector<void*> perm_c(2 * L), ext_perm(2 * L);
    char* w_perr;   // one failure flag for all lookups, read after the permuted columns' commitment (no extra sync)
    ZK_TRY(ws("cp_perr", 16, &w_perr));
    if (L) ZK_HIP(hipMemsetAsync(w_perr, 0, 16, st));
    std::vector<const void*> comp_in(L), comp_tab(L);   // the theta-compressed input / table columns (Lagrange form)
    for (uint32_t i = 0; i < L; ++i) {
        const int32_t in_col = pk->lookup_input_advice_column ? pk->lookup_input_advice_column[i] : -1;
        const int32_t tab_col = pk->lookup_table_fixed_column ? pk->lookup_table_fixed_column[i] : -1;
        for (int side = 0; side < 2; ++side) {
            if (side == 0 && in_col >= 0 && (uint32_t)in_col < A) { comp_in[i] = d_advice[in_col]; continue; }     // a single column: itself
            if (side == 1 && tab_col >= 0 && (uint32_t)tab_col < F) { comp_tab[i] = pk->fixed_lagrange[tab_col]; continue; }
            zk_evalh_args a;
            memset(&a, 0, sizeof a);
            a.k = k; a.extended_k = k; a.cs_degree = 3; a.blinding_factors = 0;
            memcpy(a.theta, theta, 32);
            a.n_fixed = F; a.n_advice = A; a.n_instance = I;
            a.n_challenges = NC; a.challenges = user_ch.data();
            a.fixed_cosets = (const uint64_t* const*)pk->fixed_lagrange;
            a.advice_cosets = (const uint64_t* const*)d_advice;
            a.instance_cosets = (const uint64_t* const*)d_instance;
            a.custom_gates = side ? pk->lookup_table_compress[i] : pk->lookup_input_compress[i];
            ZK_TRY(zkhip_evaluate_h_device(ctx, &a, w_comp + (2 * i + side) * NB));
            (side ? comp_tab[i] : comp_in[i]) = w_comp + (2 * i + side) * NB;
        }
        const void* sorted_keys = nullptr;
        if (pk->key_id && tab_col >= 0 && (uint32_t)tab_col < F)
            ZK_TRY(zk::permute_sorted_table_keys(ctx, pk->key_id, i, k, bf, comp_tab[i], &sorted_keys));
        char* bi = w_blind + (2 * i) * (bf + 1) * 32;
        char* bt = w_blind + (2 * i + 1) * (bf + 1) * 32;
        ZK_TRY(blind_rows(bl ? bl->lookup_permuted : nullptr, (size_t)(2 * i) * (bf + 1), bf + 1, bi, blinding_seed + 300 + i));
        ZK_TRY(blind_rows(bl ? bl->lookup_permuted : nullptr, (size_t)(2 * i + 1) * (bf + 1), bf + 1, bt, blinding_seed + 320 + i));
        ZK_TRY(zk::permute_expression_pair_async(ctx, k, bf, comp_in[i], comp_tab[i], bi, bt, w_perm_l + i * NB, w_perm_l + (L + i) * NB,
                                                 (uint32_t*)w_perr, sorted_keys));
    }
    for (uint32_t j = 0; j < 2 * L; ++j) { perm_c[j] = w_perm_c + j * NB; ext_perm[j] = w_ext_perm + j * EB; }
    if (L) {
        {
            std::vector<const void*> lag(2 * L);
            for (uint32_t j = 0; j < 2 * L; ++j) lag[j] = w_perm_l + j * NB;
            if (pieces_sharded) ZK_TRY(owner_intt(lag.data(), perm_c.data(), 2 * L, false, rot_perm));
            else ZK_TRY(zk::lagrange_to_coeff_oop(ctx, pk->domain, lag.data(), perm_c.data(), 2 * L));
        }
        // transcript order (lookup::Argument::commit_permuted per lookup): permuted input, then permuted table, lookup by lookup
        std::vector<const void*> cols(2 * L);
        for (uint32_t i = 0; i < L; ++i) { cols[2 * i] = perm_c[i]; cols[2 * i + 1] = perm_c[L + i]; }
        std::vector<const zkhip_srs*> bases(2 * L, pk->g);
        // the failure flag rides on the commitment's read-back: it must be known before anything enters the transcript
        uint32_t* h_err = (uint32_t*)((char*)ctx->h_pinned + zkhip_ctx::PINNED_BYTES - 16);
        ov.arm();
        ZK_TRY(zkhip_msm_g1_multi_device(ctx, bases.data(), cols.data(), cols.size(), 0, n, w_com));
        ZK_HIP(hipMemcpyAsync(h_err, w_perr, 4, hipMemcpyDeviceToHost, st));
        ZK_TRY(ov.begin_marked());
        ZK_TRY(to_extended((const void* const*)perm_c.data(), ext_perm.data(), 2 * L, rot_perm));
        ov.end();
        xy.resize(8 * cols.size());
        by.resize(32 * cols.size());
        ZK_TRY(zkhip_commitments_read(ctx, w_com, cols.size(), xy.data(), by.data()));
        if (*h_err) { set_error("permute_expression_pair: an input value is not in the table (ConstraintSystemFailure)"); return ZKHIP_ECONSTRAINT; }
        for (size_t j = 0; j < cols.size(); ++j) tr->write_point(tr->user, by.data() + 32 * j, xy.data() + 8 * j);
    }
    mark("permuted committed + absorbed");
    uint64_t beta[4], gamma[4];
    tr->squeeze_challenge(tr->user, beta);
    tr->squeeze_challenge(tr->user, gamma);
    mark("beta gamma");
    ctx->comm.phase = "grand products";

    // ---- 3. grand products: permutation sets, then lookups; commitments in coefficient form
    std::vector<void*> z_ptrs(Zp + L), ext_z(Zp + L);   // [perm sets..., lookups...]
    for (uint32_t j = 0; j < Zp + L; ++j) z_ptrs[j] = w_z + j * NB;
    {
        std::vector<const void*> values(P), ci(L), ct(L), pi(L), pt(L);
        for (uint32_t j = 0; j < P; ++j) {
            const uint32_t t = pk->perm_column_type[j], c = pk->perm_column_index[j];
            values[j] = t == 0 ? d_advice[c] : t == 1 ? pk->fixed_lagrange[c] : d_instance[c];
        }
        for (uint32_t i = 0; i < L; ++i) {
            ci[i] = comp_in[i]; ct[i] = comp_tab[i];
            pi[i] = w_perm_l + i * NB; pt[i] = w_perm_l + (L + i) * NB;
        }
        char* pb = w_blind + 2 * L * (bf + 1) * 32;
        char* lb = pb + (size_t)Zp * bf * 32;
        if (Zp) ZK_TRY(blind_rows(bl ? bl->perm_z : nullptr, 0, (size_t)Zp * bf, pb, blinding_seed + 340));
        if (L) ZK_TRY(blind_rows(bl ? bl->lookup_z : nullptr, 0, (size_t)L * bf, lb, blinding_seed + 360));
        if (pieces_sharded)   // this rank's rows of every z; the running products are completed across the ranks (32 bytes per set and rank)
            ZK_TRY(zk::grand_products_range(ctx, k, beta, gamma, bf, values.data(), pk->sigma_lagrange, P, chunk, pb, z_ptrs.data(), L, ci.data(), ct.data(),
                                            pi.data(), pt.data(), lb, z_ptrs.data() + Zp, RK * m_rows, m_rows));
        else
        ZK_TRY(zkhip_grand_products_device(ctx, k, beta, gamma, bf, values.data(), pk->sigma_lagrange, P, chunk, pb, z_ptrs.data(), L, ci.data(),
                                           ct.data(), pi.data(), pt.data(), lb, z_ptrs.data() + Zp));
    }
    if (Zp + L) {
        // extended forms in the order the sweep wants them: lookups first, then permutation sets (as the Python schedule)
        std::vector<const void*> src(Zp + L);
        for (uint32_t i = 0; i < L; ++i) { src[i] = z_ptrs[Zp + i]; ext_z[i] = w_ext_z + i * EB; }
        for (uint32_t s_ = 0; s_ < Zp; ++s_) { src[L + s_] = z_ptrs[s_]; ext_z[L + s_] = w_ext_z + (L + s_) * EB; }
        if (pieces_sharded) {   // row ranges -> the owners (batch order = the order of the coset transforms), inverse transform there, ranges back
            std::vector<void*> zb(Zp + L);
            for (uint32_t i = 0; i < Zp + L; ++i) zb[i] = const_cast<void*>(src[i]);
            ZK_TRY(owner_intt(src.data(), zb.data(), Zp + L, true, rot_z));
        } else {
            ZK_TRY(zkhip_lagrange_to_coeff_device(ctx, pk->domain, z_ptrs.data(), Zp + L));
        }
        char* w_com_rand = w_com + (size_t)(A + 2 * L + Zp + L + qd + 1) * 96;   // the last slot of the commitment area: no batch reaches it
        if (rand_late) {
            // (the grand products and the inverse transforms above are queued on the main stream; the host blocks ~1 ms inside this MSM
            // for its plan read-back and is back long before they finish)
            ZK_HIP(hipStreamWaitEvent(ctx->aux_stream, ctx->aux_event[0], 0));
            debug_delay(ctx, ctx->aux_stream);
            ctx->stream = ctx->aux_stream;
            const void* rc[1] = {w_rand};
            const zkhip_srs* rb[1] = {pk->g};
            guard.aux_launched = true;
            const int rc_ = zkhip_msm_g1_multi_device(ctx, rb, rc, 1, 0, n, w_com_rand);
            ctx->stream = st;
            ZK_TRY(rc_);
            ZK_HIP(hipEventRecord(ctx->aux_event[1], ctx->aux_stream));
        }
        std::vector<const void*> cols(z_ptrs.begin(), z_ptrs.end());
        std::vector<const zkhip_srs*> bases(Zp + L, pk->g);
        ov.arm();
        ZK_TRY(commit_launch(cols, bases));
        ZK_TRY(ov.begin_marked());
        ZK_TRY(to_extended(src.data(), ext_z.data(), Zp + L, rot_z));
        ov.end();
        ZK_TRY(commit_read(cols.size(), 0, nullptr, nullptr));
    }
    // ---- 4. the random polynomial enters the transcript here
    mark("products committed + absorbed");
    if (rand_late) {
        char* w_com_rand = w_com + (size_t)(A + 2 * L + Zp + L + qd + 1) * 96;
        ZK_HIP(event_wait(ctx, ctx->aux_event[1]));
        rand_xy.resize(8);
        rand_by.resize(32);
        ZK_TRY(zkhip_commitments_read(ctx, w_com_rand, 1, rand_xy.data(), rand_by.data()));
    }
    tr->write_point(tr->user, rand_by.data(), rand_xy.data());
    uint64_t y[4];
    tr->squeeze_challenge(tr->user, y);
    mark("y");
    ctx->comm.phase = "quotient";

    // ---- 5. quotient: sweep over the extended coset, division by the vanishing polynomial, back to coefficients, pieces
    ZK_TRY(ov.join());
    {
        zk_evalh_args a;
        memset(&a, 0, sizeof a);
        a.k = k; a.extended_k = ek; a.cs_degree = pk->cs_degree; a.blinding_factors = bf;
        memcpy(a.extended_omega, ext_omega_abi, 32); memcpy(a.g_coset, g_coset_abi, 32); memcpy(a.delta, pk->delta, 32);
        memcpy(a.beta, beta, 32); memcpy(a.gamma, gamma, 32); memcpy(a.theta, theta, 32); memcpy(a.y, y, 32);
        a.n_fixed = F; a.n_advice = A; a.n_instance = I;
        a.n_challenges = NC; a.challenges = user_ch.data();
        a.fixed_cosets = (const uint64_t* const*)(coset_mode ? kcos->fixed.data() : pk->fixed_cosets);
        a.advice_cosets = (const uint64_t* const*)ext_ptrs.data();
        a.instance_cosets = (const uint64_t* const*)(ext_ptrs.data() + A);
        a.l0 = (const uint64_t*)(coset_mode ? kcos->l0 : pk->l0);
        a.l_last = (const uint64_t*)(coset_mode ? kcos->l_last : pk->l_last);
        a.l_active_row = (const uint64_t*)(coset_mode ? kcos->l_active : pk->l_active_row);
        a.custom_gates = pk->custom_gates;
        a.n_perm_columns = P; a.n_perm_sets = Zp;
        a.perm_column_type = pk->perm_column_type; a.perm_column_index = pk->perm_column_index;
        a.perm_sigma_cosets = (const uint64_t* const*)(coset_mode ? kcos->sigma.data() : pk->sigma_cosets);
        a.perm_product_cosets = (const uint64_t* const*)(ext_z.data() + L);
        a.n_lookups = L;
        a.lookup_graphs = pk->lookup_graphs;
        a.lookup_product_cosets = (const uint64_t* const*)ext_z.data();
        a.lookup_input_cosets = (const uint64_t* const*)ext_perm.data();
        a.lookup_table_cosets = (const uint64_t* const*)(ext_perm.data() + L);
        zk::SweepCosets sc;
        if (coset_mode) zk::coset_sweep_view(cplan, &sc);
        char* vals = coset_mode ? w_hvals : w_h;
        auto sweep = [&](size_t first, size_t rows, char* dst) -> int {
            return coset_mode ? zk::evaluate_h_cosets(ctx, &a, &sc, first, rows, dst) : zkhip_evaluate_h_rows_device(ctx, &a, first, rows, dst);
        };
        if (row_mode) {
            // this rank's row range of every coset block
            for (uint32_t b_ = 0; b_ < qd; ++b_) ZK_TRY(sweep(b_ * n + RK * m_rows, m_rows, vals + (b_ * n + RK * m_rows) * 32));
            if (!pieces_sharded) {
                // the numerator's blocks are completed by one all-gather per block (the inverse transforms below run on whole blocks)
                for (uint32_t b_ = 0; b_ < qd; ++b_) ZK_TRY(zk::comm_allgather_begin(ctx, vals + (b_ * n + RK * m_rows) * 32, vals + (size_t)b_ * NB, m_rows * 32));
                ZK_TRY(zk::comm_allgather_end(ctx));
            }
        } else if (dist && ext_rows % (64 * NR) == 0) {
            const size_t rows = ext_rows / NR;
            ZK_TRY(sweep(RK * rows, rows, vals + RK * rows * 32));
            ZK_TRY(zk::comm_allgather(ctx, vals + RK * rows * 32, vals, rows * 32));
        } else {
            ZK_TRY(sweep(0, ext_rows, vals));
        }
    }
    if (pieces_sharded) {
        // Block b's inverse transform belongs to rank (b + rot_h) mod N (the round robin of the column batches goes on): (1) every rank sends its rows of block b to that owner (an all-to-all in
        // which only the owners receive), (2) the owner transforms its complete blocks, (3) sends every rank ITS row range of them, and
        // (4) each rank forms its rows of the q pieces (the q x q combination is pointwise).  2 q n / N rows cross instead of the all-gather's
        // q n, and no rank ever holds a complete piece: the commitments below read only this rank's rows.
        const size_t nb_max = (qd + NR - 1) / NR, blk = nb_max * m_rows * 32;
        char *w_send, *w_recv;
        ZK_TRY(ws("cp_a2a_send", NR * blk, &w_send));
        ZK_TRY(ws("cp_a2a_recv", NR * blk, &w_recv));
        std::vector<uint32_t> mine_blocks;
        for (uint32_t b_ = (uint32_t)first_of(rot_h, RK); b_ < qd; b_ += (uint32_t)NR) mine_blocks.push_back(b_);
        std::vector<uint8_t> owner(NR), everyone(NR, 1), nobody(NR, 0);
        for (size_t r = 0; r < NR; ++r) owner[r] = first_of(rot_h, r) < qd;
        std::vector<zk::RowCopy> list;
        const uint32_t full = 0xffffffffu;
        for (size_t r = 0; r < NR; ++r) {
            if (r == RK) continue;
            size_t t = 0;
            for (uint32_t b_ = (uint32_t)first_of(rot_h, r); b_ < qd; b_ += (uint32_t)NR, ++t)
                list.push_back(zk::RowCopy{(const uint32_t*)(w_hvals + ((size_t)b_ * n + RK * m_rows) * 32), (uint32_t*)(w_send + r * blk + t * m_rows * 32), 0u, 0u,
                                           (uint32_t)m_rows, full, full});
        }
        ZK_TRY(zk::comm_row_copies(ctx, list));
        ZK_TRY(zk::comm_alltoall(ctx, w_send, w_recv, blk, owner.data(), mine_blocks.empty() ? nobody.data() : everyone.data()));
        list.clear();
        for (size_t r = 0; r < NR && !mine_blocks.empty(); ++r) {
            if (r == RK) continue;
            for (size_t t = 0; t < mine_blocks.size(); ++t)
                list.push_back(zk::RowCopy{(const uint32_t*)(w_recv + r * blk + t * m_rows * 32), (uint32_t*)(w_hvals + ((size_t)mine_blocks[t] * n + r * m_rows) * 32), 0u,
                                           0u, (uint32_t)m_rows, full, full});
        }
        ZK_TRY(zk::comm_row_copies(ctx, list));
        ZK_TRY(zk::cosets_inverse_blocks(ctx, cplan, w_hvals, mine_blocks.data(), mine_blocks.size()));
        list.clear();
        for (size_t r = 0; r < NR && !mine_blocks.empty(); ++r) {
            if (r == RK) continue;
            for (size_t t = 0; t < mine_blocks.size(); ++t)
                list.push_back(zk::RowCopy{(const uint32_t*)(w_hvals + ((size_t)mine_blocks[t] * n + r * m_rows) * 32), (uint32_t*)(w_send + r * blk + t * m_rows * 32), 0u,
                                           0u, (uint32_t)m_rows, full, full});
        }
        ZK_TRY(zk::comm_row_copies(ctx, list));
        ZK_TRY(zk::comm_alltoall(ctx, w_send, w_recv, blk, mine_blocks.empty() ? nobody.data() : everyone.data(), owner.data()));
        list.clear();
        for (size_t r = 0; r < NR; ++r) {
            if (r == RK) continue;
            size_t t = 0;
            for (uint32_t b_ = (uint32_t)first_of(rot_h, r); b_ < qd; b_ += (uint32_t)NR, ++t)
                list.push_back(zk::RowCopy{(const uint32_t*)(w_recv + r * blk + t * m_rows * 32), (uint32_t*)(w_hvals + ((size_t)b_ * n + RK * m_rows) * 32), 0u, 0u,
                                           (uint32_t)m_rows, full, full});
        }
        ZK_TRY(zk::comm_row_copies(ctx, list));
        ZK_TRY(zk::cosets_combine_range(ctx, cplan, w_hvals, w_h, RK * m_rows, m_rows));
    } else if (coset_mode) {
        // per coset: inverse transform and s_r^-t; then the q x q combination that also carries 1 / (n (s_r^n - 1)): the division by the
        // vanishing polynomial, which is constant on a coset
        ZK_TRY(zk::cosets_to_pieces(ctx, cplan, w_hvals, w_h));
    } else {
        ZK_TRY(zkhip_divide_by_vanishing_device(ctx, pk->domain, w_h));
        void* hp[1] = {w_h};
        ZK_TRY(zkhip_extended_to_coeff_device(ctx, pk->domain, hp, 1));
    }
    std::vector<const void*> pieces(qd);
    for (uint32_t i = 0; i < qd; ++i) pieces[i] = w_h + i * NB;
    {
        std::vector<const zkhip_srs*> bases(qd, pk->g);
        ZK_TRY(commit(pieces, bases, 0, nullptr, nullptr));
    }
    mark("quotient committed + absorbed");
    tr->squeeze_challenge(tr->user, ch);
    mark("x");
    ctx->comm.phase = "evaluations";
    const HF x = hf_from_abi(ch);

    // ---- 5b. evaluations at x * omega^rotation in upstream's query order; h(X) = sum_i x^(n i) h_i(X)
    {
        const HF xn = hpow(x, n);
        std::vector<uint64_t> cf(4 * qd);
        HF acc = hone();
        for (uint32_t i = 0; i < qd; ++i) { abi_of(acc, cf.data() + 4 * i); acc = hmul(acc, xn); }
        std::vector<const void*> pl(qd);
        for (uint32_t i = 0; i < qd; ++i) pl[i] = (const char*)pieces[i] + my_lo_b;
        ZK_TRY(zkhip_linear_combination_device(ctx, my_n, pl.data(), qd, cf.data(), nullptr, 0, w_hpoly + my_lo_b));
    }
    // polynomial table for the multi-open: advice, fixed, sigma, perm_z, lookup (z, a, s) per lookup, random, h
    std::vector<const void*> polys;
    const uint32_t o_adv = 0, o_fix = A, o_sig = A + F, o_pz = A + F + P, o_lk = o_pz + Zp, o_rand = o_lk + 3 * L, o_h = o_rand + 1;
    for (uint32_t j = 0; j < A; ++j) polys.push_back(coeff_ptrs[j]);
    for (uint32_t j = 0; j < F; ++j) polys.push_back(pk->fixed_coeff[j]);
    for (uint32_t j = 0; j < P; ++j) polys.push_back(pk->sigma_coeff[j]);
    for (uint32_t j = 0; j < Zp; ++j) polys.push_back(z_ptrs[j]);
    for (uint32_t i = 0; i < L; ++i) { polys.push_back(z_ptrs[Zp + i]); polys.push_back(perm_c[i]); polys.push_back(perm_c[L + i]); }
    polys.push_back(w_rand);
    polys.push_back(w_hpoly);
    std::vector<uint32_t> q_poly;
    std::vector<int32_t> q_rot;
    auto q = [&](uint32_t poly, int32_t rot) -> uint32_t { q_poly.push_back(poly); q_rot.push_back(rot); return (uint32_t)q_poly.size() - 1; };
    const int32_t last_rot = -(int32_t)(bf + 1);
    // QUERY order (what the multi-open consumes): advice, permutation products, lookups, fixed, sigma, h, random
    std::vector<uint32_t> i_adv, i_fix, i_sig, i_pz0(Zp), i_pz1(Zp), i_pzl(Zp, ~0u), i_lz0(L), i_la0(L), i_ls0(L), i_lam(L), i_lz1(L);
    for (uint32_t j = 0; j < pk->n_advice_queries; ++j) i_adv.push_back(q(o_adv + pk->advice_query_column[j], pk->advice_query_rotation[j]));
    for (uint32_t s_ = 0; s_ < Zp; ++s_) { i_pz0[s_] = q(o_pz + s_, 0); i_pz1[s_] = q(o_pz + s_, 1); }
    for (uint32_t s_ = Zp; s_-- > 1;) i_pzl[s_ - 1] = q(o_pz + s_ - 1, last_rot);
    for (uint32_t i = 0; i < L; ++i) {
        i_lz0[i] = q(o_lk + 3 * i, 0); i_la0[i] = q(o_lk + 3 * i + 1, 0); i_ls0[i] = q(o_lk + 3 * i + 2, 0);
        i_lam[i] = q(o_lk + 3 * i + 1, -1); i_lz1[i] = q(o_lk + 3 * i, 1);
    }
    for (uint32_t j = 0; j < pk->n_fixed_queries; ++j) i_fix.push_back(q(o_fix + pk->fixed_query_column[j], pk->fixed_query_rotation[j]));
    for (uint32_t j = 0; j < P; ++j) i_sig.push_back(q(o_sig + j, 0));
    q(o_h, 0);
    const uint32_t i_rand = q(o_rand, 0);
    // WRITE order (what the transcript / the proof bytes carry; plonk/prover.rs after squeezing x): advice evals, fixed evals,
    // vanishing.evaluate (random), pk.permutation.evaluate (sigma), per set z(x), z(wx)[, z(w^last x)], per lookup z(x), z(wx), a'(x),
    // a'(w^-1 x), s'(x)
    std::vector<uint32_t> w_order;
    w_order.insert(w_order.end(), i_adv.begin(), i_adv.end());
    w_order.insert(w_order.end(), i_fix.begin(), i_fix.end());
    w_order.push_back(i_rand);
    w_order.insert(w_order.end(), i_sig.begin(), i_sig.end());
    for (uint32_t s_ = 0; s_ < Zp; ++s_) {
        w_order.push_back(i_pz0[s_]); w_order.push_back(i_pz1[s_]);
        if (s_ + 1 < Zp) w_order.push_back(i_pzl[s_]);
    }
    for (uint32_t i = 0; i < L; ++i) { w_order.push_back(i_lz0[i]); w_order.push_back(i_lz1[i]); w_order.push_back(i_la0[i]); w_order.push_back(i_lam[i]); w_order.push_back(i_ls0[i]); }
    const size_t nq = q_poly.size();
    if (nq > max_q) { set_error("zkhip_create_proof: query count"); return ZKHIP_EINVAL; }
    std::vector<uint64_t> q_points(4 * nq), q_evals(4 * nq);
    bool evals_absorbed = false;
    {
        const HF omega = hf_from_abi(omega_abi);
        std::vector<std::pair<int32_t, HF>> cache;
        for (size_t i = 0; i < nq; ++i) {
            const int32_t r = q_rot[i];
            const HF* hit = nullptr;
            for (auto& c : cache) if (c.first == r) hit = &c.second;
            if (!hit) {
                cache.push_back({r, hmul(x, hpow(omega, (uint64_t)(((int64_t)r % (int64_t)n + (int64_t)n) % (int64_t)n)))});
                hit = &cache.back().second;
            }
            abi_of(*hit, q_points.data() + 4 * i);
        }
        std::vector<const void*> qp(nq);
        for (size_t i = 0; i < nq; ++i) qp[i] = polys[q_poly[i]];
        // the evaluations are the next read-back: the kernel stores them into pinned host memory when they fit
        char* ev_out = w_evals;
        const bool ev_pinned = nq * 32 + 32768 + 64 <= zkhip_ctx::PINNED_BYTES;
        if (ev_pinned) ev_out = (char*)ctx->h_pinned + 32768;
        if (dist && ((ctx->comm.row_sharded(ctx->opt) && nq >= 2 * NR) || pieces_sharded)) {   // (with sharded pieces h(X) exists as row ranges only: always this form)
            // the evaluations are independent: rank r evaluates queries r, r + NR, ... (the 32-byte results are all-gathered, padded to the
            // same count per rank, and put back in query order).  h(X) exists only as row ranges when the pieces are sharded: every rank
            // evaluates ITS rows as a polynomial of degree < m (one more slot per rank) and h(x) = sum_R x^(R m) P_R(x) is put together here.
            const size_t ih = nq - 2;                       // q(o_h, 0) above: the last query but one
            // who evaluates query i: the owner of its polynomial when that polynomial is complete on one rank only (the new columns in the
            // owner / row-range layout), else round robin
            std::vector<int> q_rank(nq);
            std::vector<size_t> n_of(NR, 0);
            for (size_t i = 0; i < nq; ++i) {
                int o = -1;
                const uint32_t pi_ = q_poly[i];
                if (pieces_sharded) {
                    if (pi_ < A) o = (int)owner_of(rot_adv, pi_);
                    else if (pi_ >= o_pz && pi_ < o_lk) o = (int)owner_of(rot_z, L + (pi_ - o_pz));      // the products batch: lookup z's first, then the sets'
                    else if (pi_ >= o_lk && pi_ < o_rand) {      // per lookup: z (products batch), a' and s' (permuted batch: inputs first, then tables)
                        const uint32_t li = (pi_ - o_lk) / 3, wh = (pi_ - o_lk) % 3;
                        o = (int)(wh == 0 ? owner_of(rot_z, li) : owner_of(rot_perm, wh == 2 ? L + li : li));
                    }
                }
                q_rank[i] = (pieces_sharded && i == ih) ? -1 : (o >= 0 ? o : (int)(i % NR));
                if (q_rank[i] >= 0) n_of[q_rank[i]] += 1;
            }
            size_t per = 1;
            for (size_t r = 0; r < NR; ++r) per = std::max(per, n_of[r] + 1);
            std::vector<const void*> mq;
            std::vector<uint64_t> mp;
            for (size_t i = 0; i < nq; ++i) {
                if (q_rank[i] != (int)RK) continue;
                mq.push_back(qp[i]);
                mp.insert(mp.end(), q_points.begin() + 4 * i, q_points.begin() + 4 * i + 4);
            }
            char* w_evg;
            ZK_TRY(ws("cp_evals_gather", NR * per * 32, &w_evg));
            char* mine = w_evg + RK * per * 32;
            ZK_HIP(hipMemsetAsync(mine, 0, per * 32, ctx->stream));
            if (!mq.empty()) ZK_TRY(zkhip_eval_polynomials_at_device(ctx, mq.data(), mq.size(), n, mp.data(), mine));
            if (pieces_sharded) {
                const void* hp[1] = {w_hpoly + my_lo_b};
                ZK_TRY(zkhip_eval_polynomials_at_device(ctx, hp, 1, my_n, q_points.data() + 4 * ih, mine + (per - 1) * 32));
            }
            ZK_TRY(zk::comm_allgather(ctx, mine, w_evg, per * 32));
            std::vector<uint64_t> all(NR * per * 4);
            ZK_TRY(zkhip_memcpy_d2h(ctx, all.data(), w_evg, NR * per * 32));
            std::vector<size_t> taken(NR, 0);
            for (size_t i = 0; i < nq; ++i) {
                if (q_rank[i] < 0) continue;
                const size_t r = (size_t)q_rank[i];
                memcpy(q_evals.data() + 4 * i, all.data() + 4 * (r * per + taken[r]++), 32);
            }
            if (pieces_sharded) {
                const HF xm = hpow(x, m_rows);
                HF acc = hzero();
                for (size_t r = NR; r-- > 0;) acc = hadd(hmul(acc, xm), hf_from_abi(all.data() + 4 * (r * per + per - 1)));
                abi_of(acc, q_evals.data() + 4 * ih);
            }
        } else {
        // A sponge transcript spends ~15 us of HOST time per absorbed pair (Poseidon: one permutation per two evaluations) — 0.8 ms for the
        // SHA shape's 110 evaluations — with the GPU idle behind it.  With many evaluations they are therefore computed in four launches in
        // the order the transcript WRITES them, and chunk c is absorbed while chunk c + 1 is still being computed.
        uint32_t C = ctx->opt.eval_chunks > 0 ? (uint32_t)std::min(ctx->opt.eval_chunks, 4) : (ev_pinned && k <= 20 && nq >= 64 ? 4u : 1u);   // measured (gpurun_out/r04r, r04s): 110 evaluations at k = 19 in four launches -0.23 ms; 24 at k = 17 in two or three: nothing (noise); k = 22: +0.5 ms
        if (!ev_pinned || nq < 2) C = 1;
        if (C > 1) {
            std::vector<uint32_t> order(w_order);
            order.push_back((uint32_t)(nq - 2));      // h(x): evaluated, not written — last
            for (uint32_t c = 0; c < C; ++c)
                if (!ctx->eval_event[c]) ZK_HIP(hipEventCreateWithFlags(&ctx->eval_event[c], hipEventDisableTiming));
            auto lo_of = [&](uint32_t c) { return (size_t)((uint64_t)nq * c / C); };   // chunk sizes differ by at most one
            // the largest chunk first sizes the evaluation scratch once (a later, larger request would reallocate behind a stream sync)
            std::vector<const void*> cq;
            std::vector<uint64_t> cp;
            for (uint32_t c = 0; c < C; ++c) {
                const size_t b0 = lo_of(c), b1 = lo_of(c + 1);
                cq.clear(); cp.clear();
                for (size_t i = b0; i < b1; ++i) {
                    cq.push_back(qp[order[i]]);
                    cp.insert(cp.end(), q_points.begin() + 4 * (size_t)order[i], q_points.begin() + 4 * (size_t)order[i] + 4);
                }
                if (c == 0) {   // (size the scratch for the largest chunk up front)
                    void* dummy;
                    const uint32_t per_ = n >= ((size_t)1 << 16) ? 32 : 8;
                    ZK_TRY(ctx->get_scratch("po_eval_part", ((nq + C - 1) / C + 1) * (size_t)div_up(n, (size_t)per_ * 256) * 32, &dummy));
                    ZK_TRY(ctx->get_scratch("po_eval_ptrs", ((nq + C - 1) / C + 1) * sizeof(void*), &dummy));
                    ZK_TRY(ctx->get_scratch("po_eval_xs", ((nq + C - 1) / C + 1) * 32, &dummy));
                }
                ZK_TRY(zkhip_eval_polynomials_at_device(ctx, cq.data(), cq.size(), n, cp.data(), ev_out + b0 * 32));
                ZK_HIP(hipEventRecord(ctx->eval_event[c], ctx->stream));
            }
            for (uint32_t c = 0; c < C; ++c) {
                ZK_HIP(event_wait(ctx, ctx->eval_event[c]));
                for (size_t i = lo_of(c); i < lo_of(c + 1); ++i) {
                    memcpy(q_evals.data() + 4 * (size_t)order[i], ev_out + i * 32, 32);
                    if (i + 1 < nq) tr->write_scalar(tr->user, q_evals.data() + 4 * (size_t)order[i]);   // (the last entry is h(x): not written)
                }
            }
            evals_absorbed = true;
        } else {
        ZK_TRY(zkhip_eval_polynomials_at_device(ctx, qp.data(), nq, n, q_points.data(), ev_out));
        if (ev_pinned) {
            ZK_HIP(stream_wait(ctx, ctx->stream));
            memcpy(q_evals.data(), ev_out, nq * 32);
        } else {
            ZK_TRY(zkhip_memcpy_d2h(ctx, q_evals.data(), w_evals, nq * 32));
        }
        }
        }
    }
    mark("evaluations read back");
    if (!evals_absorbed)
        for (uint32_t i : w_order) tr->write_scalar(tr->user, q_evals.data() + 4 * (size_t)i);   // h(x) is not written: the verifier recomputes it
    mark("evaluations absorbed");
    ctx->comm.phase = "shplonk";
    if (out) {
        out->d_h = w_h;
        out->n_evals = nq;
        if (out->evals && out->evals_cap >= nq) memcpy(out->evals, q_evals.data(), nq * 32);
        if (out->eval_poly && out->evals_cap >= nq) memcpy(out->eval_poly, q_poly.data(), nq * 4);
        if (out->eval_rotation && out->evals_cap >= nq) memcpy(out->eval_rotation, q_rot.data(), nq * 4);
        if (out->eval_write_order && out->evals_cap >= nq) memcpy(out->eval_write_order, w_order.data(), w_order.size() * 4);
    }
    // ---- 6. SHPLONK multi-open of all of them.  The cosets of the advice / product columns are dead since the sweep: their blocks serve as
    // the multi-open's quotient / numerator scratch (2.4 GiB at k = 22 that are then never allocated)
    struct Lend {
        zkhip_ctx* c;
        ~Lend() { c->lent.clear(); }
    } lend{ctx};
    ctx->lent["sp_quot"] = zk::Scratch{w_ext, pad(A + I) * EB};
    ctx->lent["sp_num"] = zk::Scratch{w_ext_z, pad(Zp + L) * EB};
    uint64_t h1[8], h2[8];
    ZK_TRY(zk::shplonk_open(ctx, pk->g, n, polys.data(), polys.size(), q_poly.data(), q_points.data(), q_evals.data(), nq, tr, h1, h2, pieces_sharded));
    mark("shplonk done");
    if (ctx->opt.host_timing >= 2) {   // what the context holds after this proof: scratch by name (bytes), largest first
        std::vector<std::pair<size_t, std::string>> rows;
        size_t total = 0;
        for (auto& kv : ctx->scratch) { rows.push_back({kv.second.bytes, kv.first}); total += kv.second.bytes; }
        std::sort(rows.begin(), rows.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
        fprintf(stderr, "  scratch: %zu buffers, %.2f GiB\n", rows.size(), total / 1073741824.0);
        for (auto& r : rows) if (r.first >= (64u << 20)) fprintf(stderr, "    %9.1f MiB  %s\n", r.first / 1048576.0, r.second.c_str());
        size_t ptotal = 0;
        for (auto& kv : ctx->persistent) (void)kv, ptotal += 1;
        fprintf(stderr, "  persistent buffers: %zu\n", ptotal);
    }
    if (timing) {
        for (size_t i = 1; i < marks.size(); ++i)
            fprintf(stderr, "  %8.1f us  (+%7.1f)  %s\n", std::chrono::duration<double, std::micro>(marks[i].t - marks[0].t).count(),
                    std::chrono::duration<double, std::micro>(marks[i].t - marks[i - 1].t).count(), marks[i].what);
    }
    ctx->comm.phase = "";
    guard.done = true;
    return ZKHIP_OK;
}
