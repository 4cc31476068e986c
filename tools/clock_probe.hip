// What clock does the shader array actually sustain under the prover's instruction mixes, and what does one wave instruction cost in
// REAL cycles?  Every block samples s_memtime (shader clock) and s_memrealtime (constant clock) around a dense loop.
//   hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip -o tools/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct Stamp { unsigned long long c0, c1, w0, w1; };

template <int KIND, int ILP>
__global__ void k_loop(uint32_t* out, Stamp* st, int iters) {
    uint64_t acc[ILP];
    uint32_t x[ILP];
    uint32_t a = threadIdx.x * 2654435761u + 1, b = blockIdx.x * 40503u + 7;
#pragma unroll
    for (int i = 0; i < ILP; ++i) { acc[i] = i; x[i] = a + i; }
    unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {          // v_mad_u64_u32 only
#pragma unroll
            for (int i = 0; i < ILP; ++i) acc[i] = (uint64_t)a * (uint32_t)(b + i) + acc[i];
            a += (uint32_t)acc[0];
        } else if (KIND == 1) {   // 32-bit add / xor only
#pragma unroll
            for (int i = 0; i < ILP; ++i) { x[i] = (x[i] + b) ^ (x[(i + 1) % ILP] >> 3); }
        } else {                  // 1 mad : 1 simple, the field kernels' mix
#pragma unroll
            for (int i = 0; i < ILP; ++i) { acc[i] = (uint64_t)a * (uint32_t)(b + i) + acc[i]; x[i] = (x[i] + b) ^ (uint32_t)(acc[i] >> 29); }
            a += x[0];
        }
    }
    unsigned long long c1 = clock64(), w1 = wall_clock64();
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += acc[i] + x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
    if (threadIdx.x == 0) st[blockIdx.x] = Stamp{c0, c1, w0, w1};
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    int wall_khz = 0;
    CK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0));
    int cus = prop.multiProcessorCount;
    printf("device %s CUs %d nominal clock %d kHz, constant clock %d kHz\n", prop.name, cus, prop.clockRate, wall_khz);
    uint32_t* d_out;
    Stamp* d_st;
    const int maxb = cus * 8;
    CK(hipMalloc(&d_out, (size_t)maxb * 256 * 4));
    CK(hipMalloc(&d_st, maxb * sizeof(Stamp)));
    std::vector<Stamp> st(maxb);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const char* names[3] = {"mad_u64_u32", "add/xor/shift", "mad + 3 simple"};
    const int per_iter[3] = {9, 24, 33};           // VALU instructions per loop trip, counted in the gfx950 ISA of the three loops
    for (int wpc : {8, 16, 32}) {                     // waves per CU: 8 = what the 256-register kernels run at
        for (int kind = 0; kind < 3; ++kind) {
            for (int iters : {20000, 400000, 4000000}) {
                int blocks = cus * wpc / 4;
                auto launch = [&]() {
                    if (kind == 0) hipLaunchKernelGGL((k_loop<0, 8>), dim3(blocks), dim3(256), 0, 0, d_out, d_st, iters);
                    if (kind == 1) hipLaunchKernelGGL((k_loop<1, 8>), dim3(blocks), dim3(256), 0, 0, d_out, d_st, iters);
                    if (kind == 2) hipLaunchKernelGGL((k_loop<2, 8>), dim3(blocks), dim3(256), 0, 0, d_out, d_st, iters);
                };
                launch();
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                launch();
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                CK(hipMemcpy(st.data(), d_st, blocks * sizeof(Stamp), hipMemcpyDeviceToHost));
                std::vector<double> mhz(blocks), cyc(blocks);
                for (int i = 0; i < blocks; ++i) {
                    double dc = (double)(st[i].c1 - st[i].c0), dw = (double)(st[i].w1 - st[i].w0);
                    mhz[i] = dc / dw * wall_khz / 1e3;
                    cyc[i] = dc;
                }
                std::sort(mhz.begin(), mhz.end());
                std::sort(cyc.begin(), cyc.end());
                double waves_per_simd = wpc / 4.0;
                // A SIMD interleaves waves_per_simd waves, each executing iters * per_iter instructions.  The figure to quote is the one
                // from the kernel's WALL time (HIP events) at the measured shader clock: it is what a whole launch pays.  The per-block
                // cycle count (median over blocks of clock64 deltas) only covers the span in which THAT block was resident; with more
                // blocks than fit at once (4 and 8 waves per SIMD here) blocks run in turns and the median block sees fewer competitors
                // than the launch average — that column under-reports (3.07 printed where the wall time gives 4.65).
                double cpi_block = cyc[blocks / 2] / ((double)iters * per_iter[kind] * waves_per_simd);
                double cpi_wall = (double)ms * 1e-3 * mhz[blocks / 2] * 1e6 / ((double)iters * per_iter[kind] * waves_per_simd);
                printf("%-15s waves/CU=%2d iters=%8d: %8.3f ms  shader clock median %.0f MHz (min %.0f max %.0f)  %.2f shader cycles per wave instruction per SIMD "
                       "from the launch's wall time (per-block median span: %.2f)\n",
                       names[kind], wpc, iters, ms, mhz[blocks / 2], mhz[0], mhz[blocks - 1], cpi_wall, cpi_block);
            }
        }
    }
    return 0;
}
