// Instruction-rate microbenchmarks for gfx950: sets the integer roofline DESIGN.md prices MSM against.
//   hipcc --offload-arch=gfx950 -O3 -I halo2-zkcert_amd/csrc tools/microbench.hip -o tools/microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "bn254.hpp"
using namespace zk;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int ILP>
__global__ void k_mad64(uint32_t* out, int iters) {
    uint64_t acc[ILP];
    uint32_t a = threadIdx.x * 2654435761u + 1, b = blockIdx.x * 40503u + 7;
#pragma unroll
    for (int i = 0; i < ILP; ++i) acc[i] = i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) acc[i] = (uint64_t)a * (uint32_t)(b + i) + acc[i];
        a += (uint32_t)acc[0];
    }
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);
}
template <int ILP>
__global__ void k_mul24(uint32_t* out, int iters) {
    uint32_t acc[ILP];
    uint32_t a = (threadIdx.x * 2654435761u + 1) & 0xffffff, b = (blockIdx.x * 40503u + 7) & 0xffffff;
#pragma unroll
    for (int i = 0; i < ILP; ++i) acc[i] = i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) acc[i] = __umul24(a, (acc[i] + b) & 0xffffff) + acc[i];
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP>
__global__ void k_fma64(double* out, int iters) {
    double acc[ILP];
    double a = 1.0 + threadIdx.x * 1e-9, b = 1e-12 * blockIdx.x;
#pragma unroll
    for (int i = 0; i < ILP; ++i) acc[i] = i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) acc[i] = __builtin_fma(acc[i], a, b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// the field / curve level on the CURRENT field layer (typed lazy elements): the same kernels as tools/batch_affine_bench.hip
__device__ __forceinline__ el2<Fq> seeded(uint64_t seed, uint64_t gid) { return reduce(el<Fq, 32 * U>(fe_split<5>(synth_raw253(seed, gid)))); }
__global__ void k_femul(uint32_t* out, int iters, uint64_t seed) {
    uint64_t gid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    el2<Fq> x = seeded(seed, gid), y = seeded(seed + 1, gid);
    for (int it = 0; it < iters; ++it) { x = x * y; y = y * x; }
    store_raw<Fq>(out + gid * 8, x + y);
}
__global__ void k_feadd(uint32_t* out, int iters, uint64_t seed) {
    uint64_t gid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    el2<Fq> x = seeded(seed, gid), y = seeded(seed + 1, gid);
    for (int it = 0; it < iters; ++it) {   // a lazy sum and a lazy difference, contracted by conditional subtractions
        x = el2<Fq>(canonical(x + y));
        y = el2<Fq>(canonical(y - x));
    }
    store_raw<Fq>(out + gid * 8, x + y);
}
__global__ void k_madd(uint32_t* out, int iters, uint64_t seed) {
    uint64_t gid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    g1x acc;
    acc.x = seeded(seed, gid); acc.y = seeded(seed + 1, gid); acc.zz = seeded(seed + 2, gid); acc.zzz = seeded(seed + 3, gid);
    g1a q;
    q.x = seeded(seed + 4, gid); q.y = seeded(seed + 5, gid);
    for (int it = 0; it < iters; ++it) { acc = g1x_add_mixed(acc, q); q.x = reduce(acc.y + q.x); }
    store_raw<Fq>(out + gid * 8, acc.x + acc.zz);
}

template <class F>
static double time_ms(F launch, int reps = 5) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    printf("device %s CUs %d clock %d kHz\n", prop.name, cus, prop.clockRate);
    uint32_t* out; CK(hipMalloc(&out, 64ull << 20));
    const int iters = 2000;
    for (int wpc : {4, 8, 16, 32}) {  // waves per CU
        int blocks = cus * wpc / 4, threads = 256;
        double lanes = (double)blocks * threads;
        double ms = time_ms([&] { k_mad64<8><<<blocks, threads>>>(out, iters); });
        printf("mad_u64_u32 ILP8 waves/CU=%2d: %.3f ms  %.2f Tmad/s  (%.2f cycles/wave-instr/SIMD @2.4GHz)\n", wpc, ms,
               lanes * iters * 8 / ms / 1e9, ms * 1e-3 * 2.4e9 / (iters * 8.0 * wpc / 4.0));
        ms = time_ms([&] { k_mul24<8><<<blocks, threads>>>(out, iters); });
        printf("mul_u24+add ILP8 waves/CU=%2d: %.3f ms  %.2f Tmul/s\n", wpc, ms, lanes * iters * 8 / ms / 1e9);
        ms = time_ms([&] { k_fma64<8><<<blocks, threads>>>((double*)out, iters); });
        printf("fma_f64     ILP8 waves/CU=%2d: %.3f ms  %.2f Tfma/s\n", wpc, ms, lanes * iters * 8 / ms / 1e9);
        ms = time_ms([&] { k_femul<<<blocks, threads>>>(out, 500, 1); });
        printf("fe_mul      waves/CU=%2d: %.3f ms  %.2f Gmul/s\n", wpc, ms, lanes * 1000 / ms / 1e6);
        ms = time_ms([&] { k_feadd<<<blocks, threads>>>(out, 500, 1); });
        printf("fe_add/sub  waves/CU=%2d: %.3f ms  %.2f Gop/s\n", wpc, ms, lanes * 1000 / ms / 1e6);
        ms = time_ms([&] { k_madd<<<blocks, threads>>>(out, 100, 1); });
        printf("xyzz madd   waves/CU=%2d: %.3f ms  %.2f Gadd/s\n", wpc, ms, lanes * 100 / ms / 1e6);
    }
    return 0;
}
