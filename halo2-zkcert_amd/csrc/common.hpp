// common.hpp — context, error plumbing and scratch memory shared by the libzkhip.so translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <string>
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

// Named scratch buffers owned by the context, grown on demand and reused across calls so the hot
// path never calls hipMalloc (MI355X has 288 GB: scratch is sized for the largest call seen).
struct Scratch {
    void* ptr = nullptr;
    size_t bytes = 0;
};

}  // namespace zk

struct zkhip_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::map<std::string, zk::Scratch> scratch;
    // twiddle tables keyed by (log_n, omega limbs)
    struct Twiddle {
        uint32_t log_n;
        uint64_t omega[4];
        void* d_table;  // n/2 Fr
    };
    std::vector<Twiddle> twiddles;

    // optional per-kernel HIP-event timing on the launch stream (zkhip_profile_*)
    bool prof_on = false;
    struct ProfSpan { const char* name; hipEvent_t e0, e1; };
    std::vector<ProfSpan> prof_spans;
    std::vector<hipEvent_t> prof_pool;
    hipEvent_t prof_event();
    void prof_begin(const char* name);
    void prof_end();

    int get_scratch(const char* name, size_t bytes, void** out);
    int get_twiddles(const uint64_t omega[4], uint32_t log_n, const void** d_table);
};

// RAII span around one kernel launch (no-op unless profiling is enabled)
struct ProfScope {
    zkhip_ctx* c;
    ProfScope(zkhip_ctx* ctx, const char* name) : c(ctx) { if (c->prof_on) c->prof_begin(name); }
    ~ProfScope() { if (c->prof_on) c->prof_end(); }
};

static inline unsigned div_up(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }
