// Ceiling measurements for VERDICT r1 item 8 (batch-affine bucket accumulation against the XYZZ mixed addition), gfx950.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I halo2-zkcert_amd/csrc tools/batch_affine_bench.hip -o tools/batch_affine_bench
// Everything runs out of registers (no table gathers, no bucket logic), i.e. each figure is an UPPER bound for a kernel built on it:
//   xyzz_madd    : the accumulator's madd-2008-s (8M + 2S), one dependent chain per lane            -> additions / s
//   affine_core  : what a batch-affine addition costs per pair once the shared inverse exists:
//                  prefix product (1M), two back-substitution products (2M), lambda (1M), x3 (1S), y3 (1M) = 5M + 1S
//   inv_fermat / inv_euclid : one field inversion per lane (the Fermat chain; the binary extended Euclid of bn254.hpp)
// profiles/r02_msm_batch_affine.md turns these into the pairs-per-inversion a batch needs and compares that with the on-chip storage.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "bn254.hpp"
using namespace zk;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ el2<Fq> seeded(uint64_t seed, uint64_t gid) { return reduce(el<Fq, 32 * U>(fe_split<5>(synth_raw253(seed, gid)))); }

__global__ void __launch_bounds__(256) k_xyzz_madd(uint32_t* out, int iters, uint64_t seed) {
    uint64_t gid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    g1x acc;
    acc.x = seeded(seed, gid); acc.y = seeded(seed + 1, gid); acc.zz = seeded(seed + 2, gid); acc.zzz = seeded(seed + 3, gid);
    g1a q;
    q.x = seeded(seed + 4, gid); q.y = seeded(seed + 5, gid);
    for (int it = 0; it < iters; ++it) {
        acc = g1x_add_mixed(acc, q);
        q.x = reduce(acc.y + q.x);   // a fresh operand every step (one extra contraction: ~ +1M, charged to the result)
    }
    store_raw<Fq>(out + gid * 8, acc.x + acc.zz);
}

// One lane = one chain of `iters` pair additions (x1, y1) + (x2, y2) with the inverse of (x2 - x1) taken from a running
// "batch inverse" exactly as Montgomery's trick hands it out: inv_i = run * prefix_{i-1}, run *= d_i.
__global__ void __launch_bounds__(256) k_affine_core(uint32_t* out, int iters, uint64_t seed) {
    uint64_t gid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    el2<Fq> x1 = seeded(seed, gid), y1 = seeded(seed + 1, gid), x2 = seeded(seed + 2, gid), y2 = seeded(seed + 3, gid);
    el2<Fq> prefix = seeded(seed + 4, gid), run = seeded(seed + 5, gid);
    for (int it = 0; it < iters; ++it) {
        auto d = x2 - x1;
        el2<Fq> prefix_next = prefix * d;             // forward pass: 1M
        el2<Fq> inv_d = run * prefix;                 // back-substitution: 2M
        run = run * d;
        auto lambda = (y2 - y1) * inv_d;              // 1M
        auto x3 = sqr(lambda) - (x1 + x2);            // 1S
        auto y3 = lambda * (x1 - x3) - y1;            // 1M
        prefix = prefix_next;
        x1 = x2; y1 = y2;                             // the next pair: (previous second point, the sum)
        x2 = reduce(x3); y2 = reduce(y3);             // contraction of the lazy sums (~ +2M, charged to the result)
    }
    store_raw<Fq>(out + gid * 8, x2 + y2 + run + prefix);
}

__global__ void __launch_bounds__(256) k_inv_fermat(uint32_t* out, int iters, uint64_t seed) {
    uint64_t gid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    el2<Fq> a = seeded(seed, gid);
    for (int it = 0; it < iters; ++it) a = reduce(inv<Fq>(a) + one<Fq>());
    store_raw<Fq>(out + gid * 8, a);
}
// the same value in every lane of a wave (the documented way to call it: data-dependent trip counts)
__global__ void __launch_bounds__(256) k_inv_euclid(uint32_t* out, int iters, uint64_t seed) {
    uint64_t gid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    el2<Fq> a = seeded(seed, gid / 64);
    for (int it = 0; it < iters; ++it) a = reduce(inv_euclid<Fq>(a) + one<Fq>());
    store_raw<Fq>(out + gid * 8, a);
}
__global__ void __launch_bounds__(256) k_femul(uint32_t* out, int iters, uint64_t seed) {
    uint64_t gid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    el2<Fq> x = seeded(seed, gid), y = seeded(seed + 1, gid);
    for (int it = 0; it < iters; ++it) { x = x * y; y = y * x; }
    store_raw<Fq>(out + gid * 8, x + y);
}

template <class F>
static int timed(const char* name, F launch, double units, const char* unit) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    launch();
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0));
        launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("%-14s %9.3f ms  %10.3f G %s/s\n", name, best, units / best / 1e6, unit);
    return 0;
}

int main() {
    const unsigned blocks = 256 * 16, threads = 256;   // 16 workgroups per CU
    const double lanes = (double)blocks * threads;
    uint32_t* out;
    CK(hipMalloc(&out, (size_t)blocks * threads * 32));
    int it = 256;
    if (timed("fe_mul", [&] { hipLaunchKernelGGL(k_femul, dim3(blocks), dim3(threads), 0, 0, out, it, 11ull); }, lanes * it * 2, "products")) return 1;
    if (timed("xyzz_madd", [&] { hipLaunchKernelGGL(k_xyzz_madd, dim3(blocks), dim3(threads), 0, 0, out, it, 12ull); }, lanes * it, "additions")) return 1;
    if (timed("affine_core", [&] { hipLaunchKernelGGL(k_affine_core, dim3(blocks), dim3(threads), 0, 0, out, it, 13ull); }, lanes * it, "additions")) return 1;
    int it_inv = 4;
    if (timed("inv_fermat", [&] { hipLaunchKernelGGL(k_inv_fermat, dim3(blocks), dim3(threads), 0, 0, out, it_inv, 14ull); }, lanes * it_inv, "inversions")) return 1;
    if (timed("inv_euclid", [&] { hipLaunchKernelGGL(k_inv_euclid, dim3(blocks), dim3(threads), 0, 0, out, it_inv, 15ull); }, lanes * it_inv, "inversions")) return 1;
    CK(hipFree(out));
    return 0;
}
