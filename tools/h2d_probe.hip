// What does it cost to move a caller's 128 MiB Vec<Fr> (pageable host memory) to the GPU and back — the price of the host-pointer patch level
// (zkhip_msm_g1 / zkhip_fft; reference call sites /root/reference/src/helpers.rs:233,299, src/bin/cli.rs:320,369,519)?
// Measures, for one buffer size, wall-clock time and GB/s of:
//   h2d pageable   one hipMemcpyAsync from malloc'ed memory (what the round-5 zkhip_msm_g1 did)
//   h2d pinned     the same from hipHostMalloc memory (the link's rate: the floor)
//   register       hipHostRegister + chunked hipMemcpyAsync + hipHostUnregister (pin the caller's pages in place)
//   staged T       T threads memcpy chunk i+1 into a pinned ring while the DMA engine moves chunk i (T = 1, 2, 4, 8, 16)
// and the same four ways device-to-host.  Pages are touched before timing; every figure is the median of 5.
//   hipcc --offload-arch=gfx950 -O3 tools/h2d_probe.hip -o tools/h2d_probe -lpthread
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
using clk = std::chrono::steady_clock;
static double ms_since(clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }

static void par_memcpy(char* dst, const char* src, size_t bytes, int T) {
    if (T <= 1) { memcpy(dst, src, bytes); return; }
    std::vector<std::thread> th;
    const size_t per = (bytes / T + 4095) & ~(size_t)4095;
    for (int t = 0; t < T; ++t) {
        const size_t lo = std::min(bytes, per * t), hi = std::min(bytes, per * (t + 1));
        if (hi > lo) th.emplace_back([=] { memcpy(dst + lo, src + lo, hi - lo); });
    }
    for (auto& x : th) x.join();
}

int main(int argc, char** argv) {
    const size_t bytes = (argc > 1 ? (size_t)atol(argv[1]) : 128) << 20;
    const size_t chunk = (argc > 2 ? (size_t)atol(argv[2]) : 8) << 20;
    const int SLOTS = 4;
    char* pageable = (char*)aligned_alloc(4096, bytes);
    char* back = (char*)aligned_alloc(4096, bytes);
    memset(pageable, 0x5a, bytes);
    memset(back, 0, bytes);
    char *pinned, *ring, *dev;
    CK(hipHostMalloc((void**)&pinned, bytes, hipHostMallocDefault));
    CK(hipHostMalloc((void**)&ring, chunk * SLOTS, hipHostMallocDefault));
    CK(hipMalloc((void**)&dev, bytes));
    memset(pinned, 0x5a, bytes);
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t ev[SLOTS];
    for (auto& evt : ev) CK(hipEventCreateWithFlags(&evt, hipEventDisableTiming));
    printf("buffer %zu MiB, chunks of %zu MiB, ring of %d slots, %u host threads visible\n", bytes >> 20, chunk >> 20, SLOTS, std::thread::hardware_concurrency());
    auto report = [&](const char* what, std::vector<double>& v) {
        std::sort(v.begin(), v.end());
        const double med = v[v.size() / 2];
        printf("  %-28s median %8.3f ms  %7.2f GB/s   (min %.3f max %.3f)\n", what, med, bytes / med / 1e6, v.front(), v.back());
    };
    for (int dir = 0; dir < 2; ++dir) {   // 0: host -> device, 1: device -> host
        printf("%s\n", dir == 0 ? "host -> device" : "device -> host");
        std::vector<double> v;
        for (int r = 0; r < 6; ++r) {
            auto t0 = clk::now();
            if (dir == 0) CK(hipMemcpyAsync(dev, pageable, bytes, hipMemcpyHostToDevice, st)); else CK(hipMemcpyAsync(back, dev, bytes, hipMemcpyDeviceToHost, st));
            CK(hipStreamSynchronize(st));
            if (r) v.push_back(ms_since(t0));
        }
        report("pageable, one call", v);
        v.clear();
        for (int r = 0; r < 6; ++r) {
            auto t0 = clk::now();
            if (dir == 0) CK(hipMemcpyAsync(dev, pinned, bytes, hipMemcpyHostToDevice, st)); else CK(hipMemcpyAsync(pinned, dev, bytes, hipMemcpyDeviceToHost, st));
            CK(hipStreamSynchronize(st));
            if (r) v.push_back(ms_since(t0));
        }
        report("pinned, one call", v);
        v.clear();
        std::vector<double> vreg, vun;
        for (int r = 0; r < 6; ++r) {
            char* h = dir == 0 ? pageable : back;
            auto t0 = clk::now();
            CK(hipHostRegister(h, bytes, hipHostRegisterDefault));
            const double treg = ms_since(t0);
            for (size_t off = 0; off < bytes; off += chunk) {
                const size_t len = std::min(chunk, bytes - off);
                if (dir == 0) CK(hipMemcpyAsync(dev + off, h + off, len, hipMemcpyHostToDevice, st)); else CK(hipMemcpyAsync(h + off, dev + off, len, hipMemcpyDeviceToHost, st));
            }
            CK(hipStreamSynchronize(st));
            auto t1 = clk::now();
            CK(hipHostUnregister(h));
            const double tun = ms_since(t1);
            if (r) { v.push_back(ms_since(t0)); vreg.push_back(treg); vun.push_back(tun); }
        }
        report("register + copy + unregister", v);
        report("  of which hipHostRegister", vreg);
        report("  of which hipHostUnregister", vun);
        for (int T : {1, 2, 4, 8, 16}) {
            v.clear();
            for (int r = 0; r < 6; ++r) {
                auto t0 = clk::now();
                size_t i = 0;
                if (dir == 0) {
                    for (size_t off = 0; off < bytes; off += chunk, ++i) {
                        const size_t len = std::min(chunk, bytes - off);
                        const int s = (int)(i % SLOTS);
                        if (i >= (size_t)SLOTS) CK(hipEventSynchronize(ev[s]));       // the DMA that last read this slot
                        par_memcpy(ring + (size_t)s * chunk, pageable + off, len, T);
                        CK(hipMemcpyAsync(dev + off, ring + (size_t)s * chunk, len, hipMemcpyHostToDevice, st));
                        CK(hipEventRecord(ev[s], st));
                    }
                    CK(hipStreamSynchronize(st));
                } else {
                    // DMA chunk i into slot i % SLOTS; as soon as it has landed the threads copy it out while the next DMAs run
                    const size_t nch = (bytes + chunk - 1) / chunk;
                    size_t issued = 0;
                    for (size_t c = 0; c < nch; ++c) {
                        while (issued < nch && issued < c + SLOTS) {
                            const size_t off = issued * chunk, len = std::min(chunk, bytes - off);
                            CK(hipMemcpyAsync(ring + (issued % SLOTS) * chunk, dev + off, len, hipMemcpyDeviceToHost, st));
                            CK(hipEventRecord(ev[issued % SLOTS], st));
                            ++issued;
                        }
                        CK(hipEventSynchronize(ev[c % SLOTS]));
                        const size_t off = c * chunk, len = std::min(chunk, bytes - off);
                        par_memcpy(back + off, ring + (c % SLOTS) * chunk, len, T);
                    }
                }
                if (r) v.push_back(ms_since(t0));
            }
            char name[64];
            snprintf(name, sizeof name, "staged ring, %d thread%s", T, T > 1 ? "s" : "");
            report(name, v);
        }
        if (dir == 1 && memcmp(back, pageable, bytes) != 0) { printf("MISMATCH after the round trip\n"); return 1; }
    }
    // plain host memcpy rates (what the staging threads can do without the DMA)
    for (int T : {1, 2, 4, 8, 16}) {
        std::vector<double> v;
        for (int r = 0; r < 6; ++r) { auto t0 = clk::now(); par_memcpy(pinned, pageable, bytes, T); if (r) v.push_back(ms_since(t0)); }
        char name[64];
        snprintf(name, sizeof name, "host memcpy -> pinned, T=%d", T);
        report(name, v);
    }
    return 0;
}
