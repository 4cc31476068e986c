// ctx.hip — context lifetime, memory helpers, timers, host-side point helpers, synthetic fill.
#include <stdarg.h>

#include "common.hpp"
#include "hostfield.hpp"

#include <chrono>
#include <thread>

namespace zk {
static thread_local char g_err[768] = "";
static thread_local char g_stuck[256] = "";   // what a timed-out wait found (appended to the error the failing call reports)
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    int len = vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    if (g_stuck[0] && len >= 0 && (size_t)len < sizeof g_err) {
        snprintf(g_err + len, sizeof g_err - (size_t)len, " [%s]", g_stuck);
        g_stuck[0] = 0;
    }
}

AllocStats g_alloc;
// ZKHIP_POISON=1 (a test aid, read once): every fresh device allocation of the library is filled with 0xA5 bytes.  hipMalloc hands out whatever the
// previous owner — often ANOTHER process on a shared GPU — left there, while a lone process mostly sees zeros: a kernel that depends on a buffer being
// zero without clearing it passes every single-process test and fails once in a while next to other processes.  The poisoned run makes that deterministic
// (tests/test_gpu_stress.py; round 6: one unexplained digest mismatch in a six-process run).
static bool poison_on() { static const bool on = [] { const char* v = getenv("ZKHIP_POISON"); return v && atoi(v) != 0; }(); return on; }
hipError_t dev_malloc(void** p, size_t bytes) {
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipSuccess && poison_on()) { e = hipMemset(*p, 0xA5, bytes); if (e == hipSuccess) e = hipDeviceSynchronize(); }
    g_alloc.ns.fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(), std::memory_order_relaxed);
    g_alloc.bytes.fetch_add(bytes, std::memory_order_relaxed);
    g_alloc.calls.fetch_add(1, std::memory_order_relaxed);
    return e;
}

hipError_t wait_poll(const zkhip_ctx* c, hipStream_t st, hipEvent_t ev) {
    const double limit_ms = c && c->comm.nranks > 1 && c->opt.comm_timeout_ms > 0 ? (double)c->opt.comm_timeout_ms : 0.0;
    const auto t0 = std::chrono::steady_clock::now();
    if (c && (c->comm.stuck || c->dead)) {   // an earlier wait already gave up on this communicator: nothing queued behind it will ever run (dead: the flag that survives zkhip_comm_destroy)
        hipError_t e = ev ? hipEventQuery(ev) : hipStreamQuery(st);
        if (e != hipErrorNotReady) return e;
        (void)hipGetLastError();
        snprintf(g_stuck, sizeof g_stuck, "rank %d of %d: an earlier host wait of this context exceeded comm_timeout_ms; the communicator is taken for dead", c->comm.rank, c->comm.nranks);
        return hipErrorLaunchTimeOut;
    }
    for (;;) {
        for (int i = 0; i < 64; ++i) {
            hipError_t e = ev ? hipEventQuery(ev) : hipStreamQuery(st);      // (st may be the null stream: a context bound to the caller's default stream)
            if (e != hipErrorNotReady) return e;
        }
        (void)hipGetLastError();
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (limit_ms > 0.0) {
            if (ms > limit_ms) {
                snprintf(g_stuck, sizeof g_stuck, "rank %d of %d stuck after collective #%llu, phase '%s': a host wait exceeded %d ms (comm_timeout_ms)",
                         c->comm.rank, c->comm.nranks, (unsigned long long)c->comm.collectives, c->comm.phase ? c->comm.phase : "", c->opt.comm_timeout_ms);
                c->comm.stuck = 1;
                c->dead = 1;
                fprintf(stderr, "zkhip: %s\n", g_stuck);
                return hipErrorLaunchTimeOut;
            }
            if (ms > 200.0) std::this_thread::sleep_for(std::chrono::microseconds(100));   // a wait this long is not a Fiat-Shamir round trip: stop burning the core
        } else if (ms > 10000.0) {
            return ev ? hipEventSynchronize(ev) : hipStreamSynchronize(st);
        }
    }
}
}  // namespace zk
using namespace zk;

int zkhip_ctx::get_scratch(const char* name, size_t bytes, void** out) {
    // scratch is private to the stream it is used on: two streams may run the same kind of call concurrently
    char key[96];
    snprintf(key, sizeof key, "%s%s@%p", name, scratch_tag.c_str(), (void*)stream);
    if (!lent.empty()) {
        auto it = lent.find(name);
        if (it != lent.end() && it->second.bytes >= bytes && scratch.find(key) == scratch.end()) { *out = it->second.ptr; return ZKHIP_OK; }
    }
    Scratch& s = scratch[key];
    if (s.bytes < bytes) {
        if (s.ptr) {
            ZK_HIP(hipStreamSynchronize(stream));
            ZK_HIP(hipFree(s.ptr));
            s.ptr = nullptr;
            s.bytes = 0;
        }
        // a little slack on small buffers (sizes that creep up call after call); none on large ones: every GiB a context holds is
        // first-touched once — 4 ms on a warm box, ~27 ms on a freshly leased one (profiles/r04_cold_start.md)
        size_t want = bytes < ((size_t)64 << 20) ? bytes + bytes / 8 : bytes;
        hipError_t e = zk::dev_malloc((void**)&s.ptr, want);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            s.ptr = nullptr;
            set_error("hipMalloc(%zu) for scratch '%s' failed: %s", want, name, hipGetErrorString(e));
            return ZKHIP_ENOMEM;
        }
        s.bytes = want;
    }
    *out = s.ptr;
    return ZKHIP_OK;
}

int zkhip_ctx::upload(void* d_dst, const void* src, size_t bytes) {
    if (!bytes) return ZKHIP_OK;
    if (bytes > STAGE_BYTES / 4) {   // too large to stage: the classic blocking form
        ZK_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, stream));
        ZK_HIP(hipStreamSynchronize(stream));
        return ZKHIP_OK;
    }
    if (!stage_ring) ZK_HIP(hipHostMalloc(&stage_ring, STAGE_BYTES, hipHostMallocDefault));
    size_t at = (stage_off + 63) & ~(size_t)63;
    if (at + bytes > STAGE_BYTES) {
        ZK_HIP(hipDeviceSynchronize());   // every earlier staged copy has been consumed
        at = 0;
    }
    memcpy((char*)stage_ring + at, src, bytes);
    stage_off = at + bytes;
    ZK_HIP(hipMemcpyAsync(d_dst, (char*)stage_ring + at, bytes, hipMemcpyHostToDevice, stream));
    return ZKHIP_OK;
}

hipEvent_t zkhip_ctx::prof_event() {
    if (!prof_pool.empty()) { hipEvent_t e = prof_pool.back(); prof_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
void zkhip_ctx::prof_begin(const char* name) {
    ProfSpan sp{name, prof_event(), prof_event()};
    (void)hipEventRecord(sp.e0, stream);
    prof_spans.push_back(sp);
}
void zkhip_ctx::prof_end() { (void)hipEventRecord(prof_spans.back().e1, stream); }

extern "C" {

const char* zkhip_last_error(void) { return g_err; }

int zkhip_profile_enable(zkhip_ctx* c, int on) {
    if (!c) { set_error("null ctx"); return ZKHIP_EINVAL; }
    ZK_HIP(hipStreamSynchronize(c->stream));
    for (auto& sp : c->prof_spans) { c->prof_pool.push_back(sp.e0); c->prof_pool.push_back(sp.e1); }
    c->prof_spans.clear();
    c->prof_on = on != 0;
    c->prof_msm_pairs = c->prof_msm_dense_pairs = 0;
    return ZKHIP_OK;
}
int zkhip_profile_counter(zkhip_ctx* c, const char* name, uint64_t* value) {
    if (!c || !name || !value) { set_error("zkhip_profile_counter: null argument"); return ZKHIP_EINVAL; }
    if (strcmp(name, "msm_pairs") == 0) *value = c->prof_msm_pairs;
    else if (strcmp(name, "msm_dense_pairs") == 0) *value = c->prof_msm_dense_pairs;
    else if (strcmp(name, "alloc_us") == 0) *value = zk::g_alloc.ns / 1000;              // process-wide: time inside hipMalloc, bytes, calls
    else if (strcmp(name, "alloc_bytes") == 0) *value = zk::g_alloc.bytes;
    else if (strcmp(name, "alloc_calls") == 0) *value = zk::g_alloc.calls;
    else if (strcmp(name, "proofs_row_sharded") == 0) *value = c->n_row_sharded;         // always counted (not only while profiling)
    else if (strcmp(name, "proofs_pieces_sharded") == 0) *value = c->n_pieces_sharded;
    else if (strcmp(name, "shplonk_row_sharded") == 0) *value = c->n_shplonk_sharded;
    else if (strcmp(name, "comm_bulk") == 0) *value = c->comm.nccl_bulk ? 1 : 0;          // the context has a bulk communicator (comm.hip)
    else if (strcmp(name, "collectives_bulk") == 0) *value = c->comm.collectives_bulk;    // exchanges issued on it so far
    else if (strcmp(name, "comm_selfcheck") == 0) *value = c->comm.selfcheck;             // what zkhip_comm_init's self-checks ran and passed (bits: zkhip.h)
    else if (strcmp(name, "ctx_dead") == 0) *value = c->dead ? 1 : 0;                     // a host wait gave up on this context (comm_timeout_ms): only zkhip_destroy is left to call
    else { set_error("zkhip_profile_counter: unknown counter '%s'", name); return ZKHIP_EINVAL; }
    return ZKHIP_OK;
}
int zkhip_profile_select(zkhip_ctx* c, const char* kernel) {
    if (!c) { set_error("null ctx"); return ZKHIP_EINVAL; }
    c->prof_only = kernel ? kernel : "";
    return ZKHIP_OK;
}
int zkhip_profile_read(zkhip_ctx* c, const char* kernel, double* total_ms, uint64_t* launches) {
    if (!c || !kernel || !total_ms || !launches) { set_error("zkhip_profile_read: null argument"); return ZKHIP_EINVAL; }
    ZK_HIP(hipStreamSynchronize(c->stream));
    double t = 0;
    uint64_t n = 0;
    for (auto& sp : c->prof_spans) {
        if (strcmp(sp.name, kernel) != 0) continue;
        float ms = 0;
        ZK_HIP(hipEventElapsedTime(&ms, sp.e0, sp.e1));
        t += ms;
        ++n;
    }
    *total_ms = t;
    *launches = n;
    return ZKHIP_OK;
}

namespace {
struct OptName { const char* env; const char* name; int zkhip_options::*field; };
const OptName OPTIONS[] = {
    {"ZKHIP_MSM_HOST_CHUNKS", "msm_host_chunks", &zkhip_options::msm_host_chunks}, {"ZKHIP_HOST_REGISTER", "host_register", &zkhip_options::host_register}, {"ZKHIP_HOST_COPY_THREAD", "host_copy_thread", &zkhip_options::host_copy_thread}, {"ZKHIP_DEBUG_DELAY_US", "debug_delay_us", &zkhip_options::debug_delay_us}, {"ZKHIP_DEBUG_DELAY_MAIN_US", "debug_delay_main_us", &zkhip_options::debug_delay_main_us}, {"ZKHIP_MSM_C", "msm_c", &zkhip_options::msm_c}, {"ZKHIP_MSM_SEG", "msm_seg", &zkhip_options::msm_seg},
    {"ZKHIP_MSM_TAILPARTS", "msm_tailparts", &zkhip_options::msm_tailparts}, {"ZKHIP_MSM_CH", "msm_ch", &zkhip_options::msm_ch}, {"ZKHIP_MSM_TAIL2", "msm_tail2", &zkhip_options::msm_tail2},
    {"ZKHIP_MSM_WIDETAIL", "msm_widetail", &zkhip_options::msm_widetail}, {"ZKHIP_MSM_ADAPTIVE_L", "msm_adaptive_l", &zkhip_options::msm_adaptive_l},
    {"ZKHIP_MSM_DEBUG", "msm_debug", &zkhip_options::msm_debug}, {"ZKHIP_SORT_HB", "sort_hb", &zkhip_options::sort_hb},
    {"ZKHIP_SORT_TILE", "sort_tile", &zkhip_options::sort_tile}, {"ZKHIP_SORT_ONE_ATOMIC", "sort_one_atomic", &zkhip_options::sort_one_atomic}, {"ZKHIP_SORT_WIDE", "sort_wide", &zkhip_options::sort_wide}, {"ZKHIP_SORT_COPIES", "sort_copies", &zkhip_options::sort_copies},
    {"ZKHIP_NTT_SMAX", "ntt_smax", &zkhip_options::ntt_smax}, {"ZKHIP_NTT_R8", "ntt_r8", &zkhip_options::ntt_r8},
    {"ZKHIP_NTT_GROUP", "ntt_group", &zkhip_options::ntt_group}, {"ZKHIP_NTT_LDS_PAD", "ntt_lds_pad", &zkhip_options::ntt_lds_pad}, {"ZKHIP_PERMUTE_RANK_SORT", "permute_rank_sort", &zkhip_options::permute_rank_sort},
    {"ZKHIP_EVAL_BYVAL", "eval_byval", &zkhip_options::eval_byval}, {"ZKHIP_LATE_OVERLAP", "late_overlap", &zkhip_options::late_overlap},
    {"ZKHIP_HOST_TIMING", "host_timing", &zkhip_options::host_timing},
    {"ZKHIP_COSET_QUOTIENT", "coset_quotient", &zkhip_options::coset_quotient}, {"ZKHIP_ROW_SHARDED", "row_sharded", &zkhip_options::row_sharded},
    {"ZKHIP_COMM_BULK", "comm_bulk", &zkhip_options::comm_bulk}, {"ZKHIP_COMM_SELFCHECK_FORCE", "comm_selfcheck_force", &zkhip_options::comm_selfcheck_force}, {"ZKHIP_COMM_TIMEOUT_MS", "comm_timeout_ms", &zkhip_options::comm_timeout_ms}, {"ZKHIP_RAND_OVERLAP", "rand_overlap", &zkhip_options::rand_overlap},
    {"ZKHIP_EVAL_CHUNKS", "eval_chunks", &zkhip_options::eval_chunks},
};
}  // namespace

int zkhip_set_option(zkhip_ctx* c, const char* name, int value) {
    if (!c || !name) { set_error("zkhip_set_option: null argument"); return ZKHIP_EINVAL; }
    for (const auto& o : OPTIONS)
        if (strcmp(o.name, name) == 0 || strcmp(o.env, name) == 0) { c->opt.*(o.field) = value; return ZKHIP_OK; }
    set_error("zkhip_set_option: unknown option '%s'", name);
    return ZKHIP_EINVAL;
}

static int init_device_state(zkhip_ctx* c) {
    ZK_HIP(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    ZK_HIP(hipEventCreate(&c->ev0));
    ZK_HIP(hipEventCreate(&c->ev1));
    ZK_HIP(hipEventCreateWithFlags(&c->ev_read, hipEventDisableTiming));
    ZK_HIP(hipHostMalloc(&c->h_pinned, zkhip_ctx::PINNED_BYTES, hipHostMallocDefault));
    return ZKHIP_OK;
}

int zkhip_init(zkhip_ctx** out, int device_id) {
    if (!out) { set_error("zkhip_init: out is NULL"); return ZKHIP_EINVAL; }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) {
        (void)hipGetLastError();
        set_error("zkhip_init: no HIP device is visible (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
        return ZKHIP_ENODEVICE;
    }
    if (device_id < 0 || device_id >= ndev) { set_error("zkhip_init: device %d out of range [0,%d)", device_id, ndev); return ZKHIP_EINVAL; }
    ZK_HIP(hipSetDevice(device_id));
    zkhip_ctx* c = new zkhip_ctx();
    c->device = device_id;
    // the tuning knobs: read here, once (a getenv per MSM was measurable against a 0.3 ms column commitment)
    for (const auto& o : OPTIONS)
        if (const char* v = getenv(o.env)) c->opt.*(o.field) = atoi(v);
    int rc = init_device_state(c);
    if (rc != ZKHIP_OK) { zkhip_destroy(c); return rc; }   // nothing half-built survives a failed init (the message is kept)
    *out = c;
    return ZKHIP_OK;
}

void zkhip_destroy(zkhip_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->dead || c->comm.stuck) {
        // A host wait of this context gave up on a collective (comm_timeout_ms): the main stream — and whatever else is fenced behind the
        // communicator — will never drain, so hipStreamSynchronize, hipFree (which waits for the device) and hipStreamDestroy would all block for
        // ever.  Abandon every device-side resource (the runtime reclaims them when the process leaves — what a caller does after this error)
        // and return: a Rust Drop / Python close() after the deadline error must come back.
        c->comm.stuck = 1;
        (void)zkhip_comm_destroy(c);     // (stuck: touches neither RCCL nor the device)
        fprintf(stderr, "zkhip_destroy: the context was given up on by a host wait (comm_timeout_ms): its device memory and streams are abandoned, not freed\n");
        delete c;
        return;
    }
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    (void)zkhip_comm_destroy(c);
    for (auto& kv : c->scratch)
        if (kv.second.ptr) (void)hipFree(kv.second.ptr);
    for (auto& kv : c->persistent)
        if (kv.second) (void)hipFree(kv.second);
    for (auto& t : c->twiddles)
        if (t.d_lo) (void)hipFree(t.d_lo);   // lo, hi and bf share one allocation
    for (auto& sp : c->prof_spans) { (void)hipEventDestroy(sp.e0); (void)hipEventDestroy(sp.e1); }
    for (auto e : c->prof_pool) (void)hipEventDestroy(e);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev_read) (void)hipEventDestroy(c->ev_read);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    if (c->side_stream) (void)hipStreamDestroy(c->side_stream);
    if (c->side_event) (void)hipEventDestroy(c->side_event);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    for (auto e : c->eval_event) if (e) (void)hipEventDestroy(e);
    if (c->aux_stream) (void)hipStreamDestroy(c->aux_stream);
    for (auto e : c->aux_event) if (e) (void)hipEventDestroy(e);
    for (auto e : c->copy_event) if (e) (void)hipEventDestroy(e);
    for (auto e : c->host_chunk_event) if (e) (void)hipEventDestroy(e);
    if (c->copy_event_rand) (void)hipEventDestroy(c->copy_event_rand);
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    if (c->stage_ring) (void)hipHostFree(c->stage_ring);
    delete c;
}

int zkhip_set_stream(zkhip_ctx* c, void* s) {
    if (!c) { set_error("null ctx"); return ZKHIP_EINVAL; }
    c->stream = (s == ZKHIP_OWN_STREAM) ? c->own_stream : (hipStream_t)s;
    return ZKHIP_OK;
}
// Scratch buffers are keyed by the stream they were used on (two streams may run the same kind of call at once), so a caller that
// hands over many short-lived streams accumulates them: this frees every scratch buffer that does not belong to the current, own or
// side stream.  Synchronises the device.
int zkhip_trim(zkhip_ctx* c) {
    if (!c) { set_error("null ctx"); return ZKHIP_EINVAL; }
    ZK_HIP(hipDeviceSynchronize());
    char keep[4][32];
    snprintf(keep[0], 32, "@%p", (void*)c->stream);
    snprintf(keep[1], 32, "@%p", (void*)c->own_stream);
    snprintf(keep[2], 32, "@%p", (void*)c->side_stream);
    snprintf(keep[3], 32, "@%p", (void*)c->aux_stream);
    for (auto it = c->scratch.begin(); it != c->scratch.end();) {
        const std::string& k = it->first;
        bool live = false;
        for (auto& kp : keep) { size_t L = strlen(kp); if (k.size() >= L && k.compare(k.size() - L, L, kp) == 0) live = true; }
        if (!live) { if (it->second.ptr) (void)hipFree(it->second.ptr); it = c->scratch.erase(it); } else ++it;
    }
    return ZKHIP_OK;
}
int zkhip_key_release(zkhip_ctx* c, uint64_t key_id) {
    if (!c) { set_error("null ctx"); return ZKHIP_EINVAL; }
    ZK_HIP(hipDeviceSynchronize());
    char a[48], b[48];
    snprintf(a, sizeof a, "pe_table_keys:%llx:", (unsigned long long)key_id);
    snprintf(b, sizeof b, "key_cosets:%llx:", (unsigned long long)key_id);
    for (auto it = c->persistent.begin(); it != c->persistent.end();) {
        const std::string& k = it->first;
        if (k.compare(0, strlen(a), a) == 0 || k.compare(0, strlen(b), b) == 0) {
            if (it->second) (void)hipFree(it->second);
            it = c->persistent.erase(it);
        } else ++it;
    }
    zk::coset_forget_key(c, key_id);
    return ZKHIP_OK;
}
int zkhip_synchronize(zkhip_ctx* c) {
    if (!c) { set_error("null ctx"); return ZKHIP_EINVAL; }
    ZK_HIP(stream_wait(c, c->stream));      // (polling, with the communicator's deadline if there is one)
    return ZKHIP_OK;
}
int zkhip_malloc(zkhip_ctx* c, size_t bytes, void** dptr) {
    if (!c || !dptr) { set_error("zkhip_malloc: bad argument"); return ZKHIP_EINVAL; }
    hipError_t e = zk::dev_malloc((void**)dptr, bytes ? bytes : 16);
    if (e != hipSuccess) { (void)hipGetLastError(); set_error("hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); return ZKHIP_ENOMEM; }
    return ZKHIP_OK;
}
int zkhip_free(zkhip_ctx* c, void* dptr) {
    if (!c) { set_error("null ctx"); return ZKHIP_EINVAL; }
    if (dptr) { ZK_HIP(hipStreamSynchronize(c->stream)); ZK_HIP(hipFree(dptr)); }
    return ZKHIP_OK;
}
int zkhip_memcpy_h2d(zkhip_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!c) { set_error("null ctx"); return ZKHIP_EINVAL; }
    ZK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    ZK_HIP(stream_wait(c, c->stream));
    return ZKHIP_OK;
}
int zkhip_memcpy_d2h(zkhip_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!c) { set_error("null ctx"); return ZKHIP_EINVAL; }
    ZK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    ZK_HIP(stream_wait(c, c->stream));
    return ZKHIP_OK;
}
int zkhip_timer_start(zkhip_ctx* c) {
    if (!c) { set_error("null ctx"); return ZKHIP_EINVAL; }
    ZK_HIP(hipEventRecord(c->ev0, c->stream));
    return ZKHIP_OK;
}
int zkhip_timer_stop_ms(zkhip_ctx* c, float* ms) {
    if (!c || !ms) { set_error("bad argument"); return ZKHIP_EINVAL; }
    ZK_HIP(hipEventRecord(c->ev1, c->stream));
    ZK_HIP(hipEventSynchronize(c->ev1));
    ZK_HIP(hipEventElapsedTime(ms, c->ev0, c->ev1));
    return ZKHIP_OK;
}

void zkhip_g1_to_affine(const uint64_t xyz[12], uint64_t out_xy[8]) {
    g1a_store_abi(out_xy, g1j_to_affine(g1j_load_abi(xyz)));
}
// n points with ONE field inversion (Montgomery's trick): the per-batch host step between the MSM and the transcript
void zkhip_g1_batch_to_affine(const uint64_t* xyz, size_t n, uint64_t* out_xy) {
    // on the critical path of every Fiat-Shamir round trip: native 4 x 64-bit Montgomery (hostfield.hpp), the ABI form as it is
    using namespace hostmont;
    const uint64_t* P = hostfq::P;
    const uint64_t INV = hostfq::INV;
    struct Q { uint64_t w[4]; };
    std::vector<Q> pre(n);
    std::vector<uint64_t> in(xyz, xyz + 12 * n);   // canonical copies (the library's own results already are)
    for (size_t i = 0; i < 3 * n; ++i) while (geq(&in[4 * i], P)) sub_mod(&in[4 * i], P);
    xyz = in.data();
    Q acc;
    memcpy(acc.w, hostfq::ONE, 32);
    auto is_id = [&](size_t i) { const uint64_t* z = xyz + 12 * i + 8; return (z[0] | z[1] | z[2] | z[3]) == 0; };
    for (size_t i = 0; i < n; ++i) {
        pre[i] = acc;
        if (!is_id(i)) mul(acc.w, acc.w, xyz + 12 * i + 8, P, INV);
    }
    Q iv;
    {
        fe32 m;
        for (int i = 0; i < 4; ++i) { m.w[2 * i] = (uint32_t)acc.w[i]; m.w[2 * i + 1] = (uint32_t)(acc.w[i] >> 32); }
        fe32 o = to_abi(inv_host<Fq>(from_abi<Fq>(m)));
        for (int i = 0; i < 4; ++i) iv.w[i] = o.w[2 * i] | ((uint64_t)o.w[2 * i + 1] << 32);
    }
    for (size_t i = n; i-- > 0;) {
        uint64_t* o = out_xy + 8 * i;
        if (is_id(i)) { memset(o, 0, 64); continue; }
        const uint64_t* pt = xyz + 12 * i;
        Q zi, zi2, zi3;
        mul(zi.w, iv.w, pre[i].w, P, INV);
        mul(iv.w, iv.w, pt + 8, P, INV);
        mul(zi2.w, zi.w, zi.w, P, INV);
        mul(zi3.w, zi2.w, zi.w, P, INV);
        mul(o, pt, zi2.w, P, INV);
        mul(o + 4, pt + 4, zi3.w, P, INV);
    }
}
int zkhip_commitments_read(zkhip_ctx* c, const void* d_xyz, size_t n, uint64_t* out_xy, uint8_t* out_bytes) {
    if (!c || (n && (!d_xyz || !out_xy))) { set_error("zkhip_commitments_read: null argument"); return ZKHIP_EINVAL; }
    if (n == 0) return ZKHIP_OK;
    std::vector<uint64_t> big;
    uint64_t* jac = (uint64_t*)c->h_pinned;
    const char* lo = (const char*)c->h_pinned;
    if ((const char*)d_xyz >= lo && (const char*)d_xyz + n * 96 <= lo + zkhip_ctx::PINNED_BYTES) {
        // the MSM wrote its results straight into the context's pinned (device-visible) host buffer: nothing to copy
        jac = (uint64_t*)d_xyz;
    } else {
        if (n * 96 > zkhip_ctx::PINNED_BYTES) { big.resize(12 * n); jac = big.data(); }
        ZK_HIP(hipMemcpyAsync(jac, d_xyz, n * 96, hipMemcpyDeviceToHost, c->stream));
    }
    ZK_HIP(stream_wait(c, c->stream));
    zkhip_g1_batch_to_affine(jac, n, out_xy);
    if (out_bytes) for (size_t i = 0; i < n; ++i) zkhip_g1_to_bytes(out_xy + 8 * i, out_bytes + 32 * i);
    return ZKHIP_OK;
}
void zkhip_g1_add(const uint64_t a[12], const uint64_t b[12], uint64_t out[12]) {
    g1j_store_abi(out, g1j_add(g1j_load_abi(a), g1j_load_abi(b)));
}
void zkhip_g1_to_bytes(const uint64_t xy[8], uint8_t out[32]) {
    fe32 x = abi_to_canonical_words<Fq>(mem_load(xy)), y = abi_to_canonical_words<Fq>(mem_load(xy + 4));
    uint32_t any = 0;
    for (int i = 0; i < 8; ++i) any |= x.w[i] | y.w[i];
    memset(out, 0, 32);
    if (!any) { out[31] |= 0x80; return; }
    memcpy(out, x.w, 32);
    out[31] |= (uint8_t)((y.w[0] & 1) << 6);
}

}  // extern "C"

__global__ void k_synth_fill(uint32_t* out, size_t n, uint64_t seed, uint64_t first) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    mem_store(out + i * 8, synth_raw253(seed, first + i));
}

// small-valued synthetic column (bit / word columns of the SHA-256 circuit, limb columns): element i is a bit with
// probability bits_per_mille / 1000, otherwise a word of word_bits bits; stored as the Montgomery limbs of that integer.
__global__ void k_synth_small(uint32_t* out, size_t n, uint64_t seed, uint64_t first, uint32_t bits_per_mille, uint32_t word_bits) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t h = splitmix64(seed + (first + i) * 0x2545F4914F6CDD1Dull);
    uint64_t hi = h >> 32;
    uint64_t v = (h & 0xFFFFFFFFull) % 1000 < bits_per_mille ? (hi & 1) : (word_bits >= 32 ? hi : (hi & ((1ull << word_bits) - 1)));
    mem_store(out + i * 8, to_abi(from_u64<Fr>(v)));
}
extern "C" int zkhip_synth_small_device(zkhip_ctx* c, void* d_out, size_t n, uint64_t seed, uint64_t first, uint32_t bits_per_mille,
                                        uint32_t word_bits) {
    if (!c || !d_out || bits_per_mille > 1000 || word_bits == 0 || word_bits > 32) { set_error("zkhip_synth_small_device: bad argument"); return ZKHIP_EINVAL; }
    if (n == 0) return ZKHIP_OK;
    hipLaunchKernelGGL(k_synth_small, dim3(div_up(n, 256)), dim3(256), 0, c->stream, (uint32_t*)d_out, n, seed, first, bits_per_mille, word_bits);
    ZK_LAUNCH_CHECK();
    return ZKHIP_OK;
}

extern "C" int zkhip_synth_fill_device(zkhip_ctx* c, void* d_out, size_t n, uint64_t seed, uint64_t first) {
    if (!c || !d_out) { set_error("zkhip_synth_fill_device: bad argument"); return ZKHIP_EINVAL; }
    if (n == 0) return ZKHIP_OK;
    hipLaunchKernelGGL(k_synth_fill, dim3(div_up(n, 256)), dim3(256), 0, c->stream, (uint32_t*)d_out, n, seed, first);
    ZK_LAUNCH_CHECK();
    return ZKHIP_OK;
}
