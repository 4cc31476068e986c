// fake_rccl.cpp — TEST INFRASTRUCTURE: a stand-in for librccl that lets N processes sharing ONE GPU drive the RCCL transport of
// libzkhip.so (csrc/comm.hip: dlopen'ed symbols, ncclCommInitRank, event-fenced ncclAllGather, grouped ncclSend / ncclRecv with silent
// pairs) — the code path real multi-GPU nodes take, which a one-GPU box cannot run against the real library (RCCL refuses two ranks per
// device).  Selected with ZKHIP_RCCL_LIB=<this .so>.
//
// Every collective is executed synchronously on the calling host thread through a POSIX shared-memory segment named by the unique id:
// wait for the stream, copy the payload device -> shm, process barrier, copy shm -> device, barrier.  Point-to-point operations are
// collected between ncclGroupStart / ncclGroupEnd and executed at the end of the group; before any byte moves, every rank publishes the
// size of each (peer) send and receive it posted and the tables are cross-checked — a send without its matching receive (or with another
// size) is reported as an ERROR by both sides instead of the hang the real library would produce.
//   hipcc --offload-arch=gfx950 -shared -fPIC -O1 tests/fake_rccl/fake_rccl.cpp -o tests/fake_rccl/libfake_rccl.so
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <sys/mman.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <vector>

// Fault injection (tests of bench.py's fallback ladder and of the library's wait deadline), ZKFAKE_RCCL_STALL="<mode>:<rank>:<nth>": on rank
// <rank>, at the <nth> collective call after ncclCommInitRank (all-gathers and send / recv groups both count; the library's init-time
// self-check is calls 1 and 2, 1 to 5 with the bulk communicator; mode suffix "-row": only while ZKHIP_ROW_SHARDED is not 0 and ZKHIP_COMM_BULK is not 0, i.e. on the
// first rung of bench.py's ladder — row-sharded exchange with the bulk communicator asked for; the second rung is row-sharded on one communicator),
//   device — the exchange completes, then a kernel that spins for ZKFAKE_RCCL_STALL_S seconds (default 30) is enqueued on the collective's
//            stream: what a collective whose peer never arrives looks like to the host (a stream that makes no progress);
//   host   — the call never returns: what a blocked RCCL host call looks like (only an outer watchdog can end it).
__global__ void k_fake_spin(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(127);
}
static int g_calls = 0;
static void maybe_stall(int rank, hipStream_t st, bool before) {
    const char* e = getenv("ZKFAKE_RCCL_STALL");
    if (!e) return;
    char mode[16] = "";
    int r = -1, nth = -1;
    if (sscanf(e, "%15[^:]:%d:%d", mode, &r, &nth) != 3 || r != rank) return;
    if (char* suffix = strstr(mode, "-row")) {      // "device-row" / "host-row": only on the first rung (row-sharded exchange + bulk communicator)
        const char* rs = getenv("ZKHIP_ROW_SHARDED");
        if (rs && strcmp(rs, "0") == 0) return;
        const char* bk = getenv("ZKHIP_COMM_BULK");
        if (bk && strcmp(bk, "0") == 0) return;
        *suffix = 0;
    }
    if (before) { ++g_calls; if (g_calls == nth && strcmp(mode, "host") == 0) for (;;) sleep(1000); return; }
    if (g_calls == nth && strcmp(mode, "device") == 0) {
        const char* s_ = getenv("ZKFAKE_RCCL_STALL_S");
        const double secs = s_ ? atof(s_) : 30.0;
        int khz = 100000;
        (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
        hipLaunchKernelGGL(k_fake_spin, dim3(1), dim3(1), 0, st, (unsigned long long)(secs * 1000.0 * khz));
        fprintf(stderr, "fake rccl: rank %d stalls its stream for %.0f s after collective call %d (ZKFAKE_RCCL_STALL)\n", rank, secs, nth);
    }
}

extern "C" {
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;   // 0 success, 5 invalid usage, 1 unhandled (hip) error
struct FakeHeader {
    pthread_barrier_t barrier;
    volatile int ready;
    int nranks;
    size_t slot;                       // bytes per (src, dst) mailbox
    uint64_t send_size[64][64];        // [src][dst] of the group being executed
    uint64_t recv_size[64][64];        // [dst][src]
};
struct ncclComm {
    int rank, nranks;
    FakeHeader* h;
    char* data;                        // nranks x nranks mailboxes
    size_t map_bytes;
    char name[64];
};
typedef ncclComm* ncclComm_t;

static const char* g_err = "no error";
const char* ncclGetErrorString(ncclResult_t r) { return r == 0 ? "success" : g_err; }

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    memset(id->internal, 0, sizeof id->internal);
    snprintf(id->internal, sizeof id->internal, "/zkfake_rccl_%d_%ld", (int)getpid(), (long)time(nullptr));
    return 0;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
    if (nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks) { g_err = "fake rccl: bad rank / nranks"; return 5; }
    const char* s = getenv("ZKFAKE_RCCL_SLOT_MB");
    const size_t slot = (size_t)(s ? atoi(s) : 8) << 20;
    const size_t bytes = sizeof(FakeHeader) + (size_t)nranks * nranks * slot;
    ncclComm* c = new ncclComm();
    c->rank = rank; c->nranks = nranks; c->map_bytes = bytes;
    snprintf(c->name, sizeof c->name, "%s", id.internal);
    int fd = -1;
    if (rank == 0) {
        fd = shm_open(c->name, O_CREAT | O_RDWR | O_TRUNC, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) { g_err = "fake rccl: shm_open / ftruncate failed"; return 1; }
    } else {
        for (int tries = 0; tries < 20000 && fd < 0; ++tries) { fd = shm_open(c->name, O_RDWR, 0600); if (fd < 0) usleep(1000); }
        if (fd < 0) { g_err = "fake rccl: the segment never appeared"; return 1; }
        for (int tries = 0; tries < 20000; ++tries) { off_t sz = lseek(fd, 0, SEEK_END); if ((size_t)sz >= bytes) break; usleep(1000); }
    }
    void* m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { g_err = "fake rccl: mmap failed"; return 1; }
    c->h = (FakeHeader*)m;
    c->data = (char*)m + sizeof(FakeHeader);
    if (rank == 0) {
        pthread_barrierattr_t a;
        pthread_barrierattr_init(&a);
        pthread_barrierattr_setpshared(&a, PTHREAD_PROCESS_SHARED);
        pthread_barrier_init(&c->h->barrier, &a, (unsigned)nranks);
        c->h->nranks = nranks;
        c->h->slot = slot;
        __sync_synchronize();
        c->h->ready = 1;
    } else {
        for (int tries = 0; tries < 20000 && !c->h->ready; ++tries) usleep(1000);
        if (!c->h->ready) { g_err = "fake rccl: rank 0 never initialised the segment"; return 1; }
    }
    pthread_barrier_wait(&c->h->barrier);
    if (rank == 0) shm_unlink(c->name);     // the mappings keep it alive
    *out = c;
    return 0;
}
// a split over the SAME ranks (color / key as the library passes them: one colour, key = rank): another segment, named after the parent's
ncclResult_t ncclCommSplit(ncclComm_t parent, int /*color*/, int key, ncclComm_t* out, void* /*config*/) {
    static int n_splits = 0;
    if (!parent || key != parent->rank) { g_err = "fake rccl: ncclCommSplit supports one colour with key = rank"; return 5; }
    ncclUniqueId id;
    memset(id.internal, 0, sizeof id.internal);
    snprintf(id.internal, sizeof id.internal, "%.40s_s%d", parent->name, ++n_splits);      // every rank splits in the same order: the same name
    return ncclCommInitRank(out, parent->nranks, id, parent->rank);
}
ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return 0;
    munmap((void*)c->h, c->map_bytes);
    delete c;
    return 0;
}
ncclResult_t ncclCommCount(const ncclComm_t c, int* n) { *n = c->nranks; return 0; }
ncclResult_t ncclCommUserRank(const ncclComm_t c, int* r) { *r = c->rank; return 0; }

static char* box(ncclComm_t c, int src, int dst) { return c->data + ((size_t)src * c->nranks + dst) * c->h->slot; }

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, int /*dtype: bytes*/, ncclComm_t c, hipStream_t st) {
    if (count > c->h->slot) { g_err = "fake rccl: all-gather block larger than ZKFAKE_RCCL_SLOT_MB"; return 5; }
    maybe_stall(c->rank, st, true);
    ncclResult_t rc = 0;      // (every path reaches both barriers: an error on one rank must not strand the others)
    if (hipStreamSynchronize(st) != hipSuccess) { g_err = "fake rccl: hipStreamSynchronize"; rc = 1; }
    if (!rc && hipMemcpy(box(c, c->rank, c->rank), send, count, hipMemcpyDeviceToHost) != hipSuccess) { g_err = "fake rccl: D2H"; rc = 1; }
    pthread_barrier_wait(&c->h->barrier);
    for (int r = 0; r < c->nranks && !rc; ++r)
        if (hipMemcpy((char*)recv + (size_t)r * count, box(c, r, r), count, hipMemcpyHostToDevice) != hipSuccess) { g_err = "fake rccl: H2D"; rc = 1; }
    pthread_barrier_wait(&c->h->barrier);
    maybe_stall(c->rank, st, false);
    return rc;
}

struct P2P { int send; const void* sbuf; void* rbuf; size_t bytes; int peer; ncclComm_t c; hipStream_t st; };
static thread_local std::vector<P2P> g_ops;
static thread_local int g_depth = 0;
static ncclResult_t run_group() {
    if (g_ops.empty()) return 0;
    ncclComm_t c = g_ops[0].c;
    const int me = c->rank, N = c->nranks;
    hipStream_t st0 = g_ops[0].st;
    maybe_stall(me, st0, true);
    for (int r = 0; r < N; ++r) { c->h->send_size[me][r] = 0; c->h->recv_size[me][r] = 0; }
    ncclResult_t rc = 0;
    for (auto& o : g_ops) {
        if (o.peer < 0 || o.peer >= N || o.peer == me || o.bytes > c->h->slot) { g_err = "fake rccl: bad peer or block larger than ZKFAKE_RCCL_SLOT_MB"; rc = 5; }
        else if (o.send) c->h->send_size[me][o.peer] += o.bytes; else c->h->recv_size[me][o.peer] += o.bytes;
    }
    if (hipStreamSynchronize(g_ops[0].st) != hipSuccess) { g_err = "fake rccl: hipStreamSynchronize"; rc = 1; }
    for (auto& o : g_ops)
        if (!rc && o.send && hipMemcpy(box(c, me, o.peer), o.sbuf, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) { g_err = "fake rccl: D2H"; rc = 1; }
    pthread_barrier_wait(&c->h->barrier);
    // every send must meet a receive of the same size, and the other way round: the real library would hang here
    for (int r = 0; r < N && !rc; ++r) {
        if (r == me) continue;
        if (c->h->send_size[me][r] != c->h->recv_size[r][me]) { g_err = "fake rccl: a send of this rank has no matching receive of that size on the peer (the real library would hang)"; rc = 5; }
        if (c->h->recv_size[me][r] != c->h->send_size[r][me]) { g_err = "fake rccl: a receive of this rank has no matching send of that size on the peer (the real library would hang)"; rc = 5; }
    }
    for (auto& o : g_ops)
        if (!rc && !o.send && hipMemcpy(o.rbuf, box(c, o.peer, me), o.bytes, hipMemcpyHostToDevice) != hipSuccess) { g_err = "fake rccl: H2D"; rc = 1; }
    pthread_barrier_wait(&c->h->barrier);
    g_ops.clear();
    maybe_stall(me, st0, false);
    return rc;
}
ncclResult_t ncclGroupStart() { ++g_depth; return 0; }
ncclResult_t ncclGroupEnd() {
    if (g_depth <= 0) { g_err = "fake rccl: ncclGroupEnd without ncclGroupStart"; return 5; }
    if (--g_depth == 0) return run_group();
    return 0;
}
ncclResult_t ncclSend(const void* buf, size_t count, int, int peer, ncclComm_t c, hipStream_t st) {
    g_ops.push_back(P2P{1, buf, nullptr, count, peer, c, st});
    return g_depth ? 0 : run_group();
}
ncclResult_t ncclRecv(void* buf, size_t count, int, int peer, ncclComm_t c, hipStream_t st) {
    g_ops.push_back(P2P{0, nullptr, buf, count, peer, c, st});
    return g_depth ? 0 : run_group();
}
}  // extern "C"
