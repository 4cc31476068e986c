// replay_rccl.cpp — MEASUREMENT INFRASTRUCTURE (not the product, not a test double for correctness): a stand-in for librccl that lets ONE
// process on ONE idle GPU run exactly what rank R of an N-rank proof runs — the kernels, their sizes, the launch structure, the event
// fences and the communicator stream of csrc/comm.hip's RCCL branch — with the peers' contributions FABRICATED on the device instead of
// received.  bench.py --replay-rank R --of N binds it through zkhip_comm_use_library; the proof that comes out is wrong by construction
// (bench.py says so in the line and never compares it with anything); what is measured is rank R's own share of the work, which a one-GPU
// box can time alone — the numbers that do not divide by N.
//
// What each entry point does (all of it asynchronous on the caller's stream, like the real library):
//   ncclAllGather   own block copied device-to-device into its slot, every peer's slot FILLED by a kernel;
//   ncclSend        nothing (the bytes would leave over this rank's outgoing links);
//   ncclRecv        the receive buffer FILLED by a kernel;
//   ncclGroupEnd    (outermost) and ncclAllGather: optionally a MODELLED wire time — a kernel that holds the communicator's stream for
//                   latency + max-over-peers(bytes received from that peer) / per-link bandwidth (xGMI is point-to-point: one link per peer,
//                   all of them busy at once) — ZKREPLAY_LATENCY_US / ZKREPLAY_LINK_GBS; both 0 (default): exchanges cost only their fill.
// Fill contents: blocks of <= 64 bytes take what csrc/comm.hip's init-time self-checks expect from a healthy peer (the word peer << 16 | me
// from a receive — 0xB0000000 | peer << 12 | me on a communicator made by ncclCommSplit, the library's bulk communicator —, the verdict word 1
// from an all-gather); larger blocks are pseudo-random 32-byte rows with the top limb masked below both
// BN254 moduli, so that scalars taken from a peer's rows spread over the MSM's buckets like real coefficients do (a constant fill would put
// every point of a window into one bucket and time a pathological accumulation).
// Counters (ncclReplayStats, read by bench.py through ctypes): collectives, bytes received, bytes sent, modelled wire microseconds.
//   hipcc --offload-arch=gfx950 -shared -fPIC -O2 tools/replay_rccl/replay_rccl.cpp -o tools/replay_rccl/libreplay_rccl.so
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

__global__ void __launch_bounds__(256) k_replay_fill(uint32_t* dst, size_t words, uint32_t small_word, uint32_t seed, int random_rows) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += stride) {
        uint32_t v = small_word;
        if (random_rows) {
            uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull + ((uint64_t)seed << 32);
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            v = (uint32_t)(z ^ (z >> 31));
            if ((i & 7) == 7) v &= 0x0FFFFFFFu;     // top limb of a 32-byte row: the value stays below r and q
        }
        dst[i] = v;
    }
}
__global__ void k_replay_wire(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

extern "C" {
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
struct ncclComm {
    int rank, nranks;
    double latency_us, link_gbs;
    int wall_khz;
    int split;      // made by ncclCommSplit (the library's bulk communicator): its init-time self-check expects other tag words
};
typedef ncclComm* ncclComm_t;
struct ReplayStats { uint64_t collectives, bytes_received, bytes_sent; double wire_us; };
static ReplayStats g_stats = {0, 0, 0, 0.0};
static const char* g_err = "no error";
const char* ncclGetErrorString(ncclResult_t r) { return r == 0 ? "success" : g_err; }
void ncclReplayStats(ReplayStats* out, int reset) { if (out) *out = g_stats; if (reset) g_stats = ReplayStats{0, 0, 0, 0.0}; }

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    memset(id->internal, 0, sizeof id->internal);
    snprintf(id->internal, sizeof id->internal, "zkreplay");
    return 0;
}
ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId, int rank) {
    if (nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks) { g_err = "replay rccl: bad rank / nranks"; return 5; }
    ncclComm* c = new ncclComm();
    c->rank = rank; c->nranks = nranks;
    const char* l = getenv("ZKREPLAY_LATENCY_US");
    const char* b = getenv("ZKREPLAY_LINK_GBS");
    c->latency_us = l ? atof(l) : 0.0;
    c->link_gbs = b ? atof(b) : 0.0;
    c->wall_khz = 100000;
    c->split = 0;
    (void)hipDeviceGetAttribute(&c->wall_khz, hipDeviceAttributeWallClockRate, 0);
    *out = c;
    return 0;
}
ncclResult_t ncclCommSplit(ncclComm_t parent, int, int, ncclComm_t* out, void*) {      // the library's bulk communicator: same rank / size / wire model, own handle
    *out = new ncclComm(*parent);
    (*out)->split = 1;
    return 0;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) { delete c; return 0; }
ncclResult_t ncclCommCount(const ncclComm_t c, int* n) { *n = c->nranks; return 0; }
ncclResult_t ncclCommUserRank(const ncclComm_t c, int* r) { *r = c->rank; return 0; }

static ncclResult_t fill(void* dst, size_t bytes, uint32_t small_word, uint32_t seed, hipStream_t st) {
    if (!bytes) return 0;
    if (bytes % 4) { g_err = "replay rccl: block size is not a multiple of 4 bytes"; return 5; }
    const size_t words = bytes / 4;
    const unsigned blocks = (unsigned)std::min<size_t>((words + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(k_replay_fill, dim3(blocks), dim3(256), 0, st, (uint32_t*)dst, words, small_word, seed, bytes > 64 ? 1 : 0);
    if (hipGetLastError() != hipSuccess) { g_err = "replay rccl: fill launch failed"; return 1; }
    return 0;
}
static ncclResult_t wire(ncclComm_t c, size_t worst_link_bytes, hipStream_t st) {
    if (c->latency_us <= 0.0 && c->link_gbs <= 0.0) return 0;
    double us = c->latency_us;
    if (c->link_gbs > 0.0) us += (double)worst_link_bytes / (c->link_gbs * 1e3);
    g_stats.wire_us += us;
    hipLaunchKernelGGL(k_replay_wire, dim3(1), dim3(1), 0, st, (unsigned long long)(us * 1e-3 * c->wall_khz));
    if (hipGetLastError() != hipSuccess) { g_err = "replay rccl: wire launch failed"; return 1; }
    return 0;
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, int, ncclComm_t c, hipStream_t st) {
    char* mine = (char*)recv + (size_t)c->rank * count;
    if ((const void*)mine != send && hipMemcpyAsync(mine, send, count, hipMemcpyDeviceToDevice, st) != hipSuccess) { g_err = "replay rccl: D2D"; return 1; }
    for (int r = 0; r < c->nranks; ++r) {
        if (r == c->rank) continue;
        if (ncclResult_t rc = fill((char*)recv + (size_t)r * count, count, 1u, 0xA60000u + (uint32_t)g_stats.collectives * 64u + (uint32_t)r, st)) return rc;
    }
    g_stats.collectives += 1;
    g_stats.bytes_received += count * (size_t)(c->nranks - 1);
    g_stats.bytes_sent += count * (size_t)(c->nranks - 1);
    return wire(c, count, st);       // every peer's block arrives over that peer's own link
}

static thread_local int g_depth = 0;
static thread_local size_t g_worst = 0;
static thread_local ncclComm_t g_comm = nullptr;
static thread_local hipStream_t g_stream = nullptr;
static ncclResult_t end_group() {
    if (!g_comm) return 0;
    g_stats.collectives += 1;
    ncclResult_t rc = wire(g_comm, g_worst, g_stream);
    g_comm = nullptr; g_worst = 0;
    return rc;
}
ncclResult_t ncclGroupStart() { ++g_depth; return 0; }
ncclResult_t ncclGroupEnd() {
    if (g_depth <= 0) { g_err = "replay rccl: ncclGroupEnd without ncclGroupStart"; return 5; }
    if (--g_depth == 0) return end_group();
    return 0;
}
ncclResult_t ncclSend(const void*, size_t count, int, int peer, ncclComm_t c, hipStream_t st) {
    if (peer < 0 || peer >= c->nranks || peer == c->rank) { g_err = "replay rccl: bad peer"; return 5; }
    g_stats.bytes_sent += count;
    g_comm = c; g_stream = st; g_worst = std::max(g_worst, count);
    return g_depth ? 0 : end_group();
}
ncclResult_t ncclRecv(void* buf, size_t count, int, int peer, ncclComm_t c, hipStream_t st) {
    if (peer < 0 || peer >= c->nranks || peer == c->rank) { g_err = "replay rccl: bad peer"; return 5; }
    const uint32_t tag = c->split ? (0xB0000000u | ((uint32_t)peer << 12) | (uint32_t)c->rank) : (((uint32_t)peer << 16) | (uint32_t)c->rank);      // what csrc/comm.hip's self-checks expect from peer
    if (ncclResult_t rc = fill(buf, count, tag, 0x5E0000u + (uint32_t)g_stats.collectives * 64u + (uint32_t)peer, st)) return rc;
    g_stats.bytes_received += count;
    g_comm = c; g_stream = st; g_worst = std::max(g_worst, count);
    return g_depth ? 0 : end_group();
}
}  // extern "C"
