// What would a Fiat-Shamir point cost if it stayed on the device?  ("Keccak / Poseidon / Blake2b absorb-squeeze as
// one-workgroup device kernels fed by the on-device to_affine results".)  A transcript step is a SEQUENTIAL computation on a handful of
// values: normalise the commitments (one field inversion for the batch), absorb them (Keccak-f / the Poseidon permutation), squeeze.
// This probe times exactly those pieces as ONE wave (<<<1, 64>>>, every lane the same work — the latency of one lane is what matters),
// with the library's own field arithmetic, against the host round trip they would replace (~50 us per Fiat-Shamir point today:
// profiles/r03_* gap traces; host Poseidon ~10 us per absorbed pair, host inversion ~1.5 us).
//   hipcc --offload-arch=gfx950 -O3 -I halo2-zkcert_amd/csrc tools/fs_device_probe.hip -o tools/fs_device_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include "bn254.hpp"
using namespace zk;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_inv_euclid(const uint32_t* in, uint32_t* out, int reps) {
    el2<Fq> a = from_abi<Fq>(mem_load(in));
    for (int i = 0; i < reps; ++i) a = reduce(inv_euclid<Fq>(a) + one<Fq>());
    mem_store(out + threadIdx.x * 8, to_abi(a));
}
__global__ void k_inv_fermat(const uint32_t* in, uint32_t* out, int reps) {
    el2<Fq> a = from_abi<Fq>(mem_load(in));
    for (int i = 0; i < reps; ++i) a = reduce(inv<Fq>(a) + one<Fq>());
    mem_store(out + threadIdx.x * 8, to_abi(a));
}
// a chain of dependent Fr products: the Poseidon permutation snark-verifier's transcript uses (t = 3, 8 full + 57 partial rounds, x^5) is
// 8 * (3 * 3 + 9) + 57 * (3 + ~5) = ~600 products deep in one lane (the three state words of a full round are independent, the rest is not)
__global__ void k_mul_chain(const uint32_t* in, uint32_t* out, int n) {
    el2<Fr> a = from_abi<Fr>(mem_load(in)), b = a;
    for (int i = 0; i < n; ++i) a = reduce(a * b + a);
    mem_store(out + threadIdx.x * 8, to_abi(a));
}
__device__ __forceinline__ uint64_t rotl64(uint64_t x, int r) { return r ? (x << r) | (x >> (64 - r)) : x; }
__global__ void k_keccak(uint64_t* st_io, int perms) {
    const uint64_t RC[24] = {0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808Aull, 0x8000000080008000ull, 0x000000000000808Bull, 0x0000000080000001ull,
                             0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008Aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000Aull,
                             0x000000008000808Bull, 0x800000000000008Bull, 0x8000000000008089ull, 0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull,
                             0x000000000000800Aull, 0x800000008000000Aull, 0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
    const int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    const int PI[25] = {0, 10, 20, 5, 15, 16, 1, 11, 21, 6, 7, 17, 2, 12, 22, 23, 8, 18, 3, 13, 14, 24, 9, 19, 4};
    uint64_t A[25];
    for (int i = 0; i < 25; ++i) A[i] = st_io[i];
    for (int p = 0; p < perms; ++p)
        for (int r = 0; r < 24; ++r) {
            uint64_t C[5], B[25];
            for (int x = 0; x < 5; ++x) C[x] = A[x] ^ A[x + 5] ^ A[x + 10] ^ A[x + 15] ^ A[x + 20];
            for (int x = 0; x < 5; ++x) { uint64_t d = C[(x + 4) % 5] ^ rotl64(C[(x + 1) % 5], 1); for (int y = 0; y < 25; y += 5) A[x + y] ^= d; }
            for (int i = 0; i < 25; ++i) B[PI[i]] = rotl64(A[i], ROT[i]);
            for (int y = 0; y < 25; y += 5) for (int x = 0; x < 5; ++x) A[x + y] = B[x + y] ^ (~B[(x + 1) % 5 + y] & B[(x + 2) % 5 + y]);
            A[0] ^= RC[r];
        }
    if (threadIdx.x == 0) for (int i = 0; i < 25; ++i) st_io[i] = A[i];
}
__global__ void k_empty() {}

int main() {
    uint32_t h[8] = {0x12345671u, 0x9abcdef0u, 0x0fedcba9u, 0x87654321u, 0x13579bdfu, 0x2468ace0u, 0x0badf00du, 0x01234567u};
    uint32_t *d_in, *d_out;
    uint64_t* d_st;
    CK(hipMalloc(&d_in, 32));
    CK(hipMalloc(&d_out, 64 * 32));
    CK(hipMalloc(&d_st, 25 * 8));
    CK(hipMemcpy(d_in, h, 32, hipMemcpyHostToDevice));
    CK(hipMemset(d_st, 1, 25 * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto time_us = [&](auto&& launch, int reps) -> double {
        launch();
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        launch();
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        return ms * 1000.0 / reps;
    };
    const int R = 16;
    printf("one wave (<<<1, 64>>>), per operation:\n");
    printf("  empty kernel (launch + event overhead)            %8.1f us\n", time_us([&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0); }, 1));
    printf("  Fq inversion, binary Euclid (inv_euclid)          %8.1f us\n", time_us([&] { hipLaunchKernelGGL(k_inv_euclid, dim3(1), dim3(64), 0, 0, d_in, d_out, R); }, R));
    printf("  Fq inversion, Fermat chain (inv)                  %8.1f us\n", time_us([&] { hipLaunchKernelGGL(k_inv_fermat, dim3(1), dim3(64), 0, 0, d_in, d_out, R); }, R));
    printf("  600 dependent Fr products (one Poseidon perm.)    %8.1f us\n", time_us([&] { hipLaunchKernelGGL(k_mul_chain, dim3(1), dim3(64), 0, 0, d_in, d_out, 600 * R); }, R));
    printf("  Keccak-f[1600], one lane                          %8.1f us\n", time_us([&] { hipLaunchKernelGGL(k_keccak, dim3(1), dim3(64), 0, 0, d_st, R); }, R));
    return 0;
}
