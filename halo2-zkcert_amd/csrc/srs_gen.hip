// srs_gen.hip — ParamsKZG::setup(k, rng) restricted to G1, on the device.
//
// Mirrors halo2_proofs poly/kzg/commitment.rs ParamsKZG::setup [UPSTREAM-RECALL; crate pinned at
// /root/reference/Cargo.lock:1320-1322; reached through gen_srs at /root/reference/src/helpers.rs:213,
// src/bin/cli.rs:306]: g[i] = [s^i] G and g_lagrange[i] = [l_i(s)] G.  Upstream multiplies the running
// point by s; the points are unique, so here each base is an independent fixed-base multiplication
// with an 8-bit windowed table of G (32 mixed additions per base).
#include <vector>

#include "common.hpp"
using namespace zk;

// table[w][d] = [d * 256^w] G, affine; d = 0 is the identity.
__global__ void k_fixed_base_table(uint32_t* table_xy) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 32 * 256) return;
    uint32_t w = t >> 8, d = t & 255;
    g1a gen;
    gen.x = from_u64<Fq>(1);
    gen.y = from_u64<Fq>(2);
    g1j base = g1j_from_affine(gen);
    for (uint32_t i = 0; i < 8 * w; ++i) base = g1j_double(base);
    g1j acc = g1j_identity();
    for (int b = 7; b >= 0; --b) {
        acc = g1j_double(acc);
        if ((d >> b) & 1) acc = g1j_add(acc, base);
    }
    g1a_store_raw(table_xy + (size_t)t * 16, g1j_to_affine(acc));
}
__global__ void __launch_bounds__(256) k_fixed_base_mul(const uint32_t* scalars, size_t n, const uint32_t* table_xy, uint32_t* out_xyz) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe32 k = abi_to_canonical_words<Fr>(mem_load(scalars + i * 8));
    g1j acc = g1j_identity();
#pragma unroll 1
    for (uint32_t w = 0; w < 32; ++w) {
        uint32_t limb = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) limb = (w >> 2) == (uint32_t)j ? k.w[j] : limb;
        uint32_t d = (limb >> ((w & 3) * 8)) & 255;
        if (d) acc = g1j_add_mixed(acc, g1a_load_raw(table_xy + ((size_t)w * 256 + d) * 16));
    }
    g1j_store_raw(out_xyz + i * 24, acc);
}
// The setup's scalars on the device (the host loop — 2^22 x ~7 products in the limb layer compiled for the host — was 5 of the 6.3 s
// that bench.py reported as setup_s at k = 22): 16 consecutive indices per thread, the first power by square-and-multiply.
//   mono:     out[i] = s^(first + i)
//   lagrange: pass 1  out[i] = s - w^(first + i)            (then one batch inversion over the array)
//             pass 2  out[i] = num * w^(first + i) * out[i]  with num = (s^n - 1) / n:   l_i(s)
// All in the ABI form (Montgomery, R = 2^256), as the fixed-base multiplication below reads them.
__global__ void k_setup_scalars(uint32_t* out, size_t count, size_t first, fe base_v, fe s_v, fe num_v, int mode) {
    const size_t CHK = 16;
    const size_t lo = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) * CHK;
    if (lo >= count) return;
    const el2<Fr> base(base_v), sv(s_v), num(num_v);
    el2<Fr> w = pow_u64<Fr>(base, (uint64_t)(first + lo));
    const size_t hi = lo + CHK < count ? lo + CHK : count;
    for (size_t i = lo; i < hi; ++i) {
        if (mode == 0) store_div32<Fr>(out + i * 8, w);
        else if (mode == 1) store_div32<Fr>(out + i * 8, reduce(sv - w));
        else store_div32<Fr>(out + i * 8, (num * w) * load_x32<Fr>(out + i * 8));
        w = w * base;
    }
}

namespace zk {  // msm.hip
int launch_batch_to_affine(zkhip_ctx* ctx, const void* d_in_xyz, void* d_out_xy, size_t n);
int srs_build_raw(zkhip_ctx* ctx, const void* d_bases_raw, size_t n, zkhip_srs** out);
}

extern "C" int zkhip_fixed_base_mul_device(zkhip_ctx* ctx, const void* d_scalars, size_t n, void* d_out_xy);

// device scalars (ABI Montgomery Fr) -> affine bases in the library's table form (R' = 2^261, canonical)
extern "C" int zkhip_fixed_base_mul_device(zkhip_ctx* ctx, const void* d_scalars, size_t n, void* d_out_xy) {
    if (!ctx || !d_scalars || !d_out_xy) { set_error("zkhip_fixed_base_mul_device: null argument"); return ZKHIP_EINVAL; }
    if (n == 0) return ZKHIP_OK;
    void *d_table, *d_jac;
    char key[96];
    snprintf(key, sizeof key, "fixed_base_table@%p", (void*)ctx->stream);
    auto it = ctx->scratch.find(key);
    bool have = it != ctx->scratch.end() && it->second.ptr;
    ZK_TRY(ctx->get_scratch("fixed_base_table", 32 * 256 * 64, &d_table));
    ZK_TRY(ctx->get_scratch("fixed_base_jac", n * 96, &d_jac));
    hipStream_t st = ctx->stream;
    if (!have) hipLaunchKernelGGL(k_fixed_base_table, dim3(32 * 256 / 64), dim3(64), 0, st, (uint32_t*)d_table);
    hipLaunchKernelGGL(k_fixed_base_mul, dim3(div_up(n, 256)), dim3(256), 0, st, (const uint32_t*)d_scalars, n, (const uint32_t*)d_table,
                       (uint32_t*)d_jac);
    ZK_LAUNCH_CHECK();
    return launch_batch_to_affine(ctx, d_jac, d_out_xy, n);
}

namespace zk { int srs_set_range(zkhip_srs* s, size_t first, size_t n_total); }

extern "C" int zkhip_kzg_setup_range(zkhip_ctx* ctx, uint32_t k, const uint64_t s_u[4], size_t first, size_t count, zkhip_srs** g,
                                     zkhip_srs** g_lagrange) {
    if (!ctx || !s_u || (!g && !g_lagrange)) { set_error("zkhip_kzg_setup: null argument"); return ZKHIP_EINVAL; }
    if (k > 24) { set_error("zkhip_kzg_setup: k = %u unsupported (max 24)", k); return ZKHIP_EINVAL; }
    const size_t n = (size_t)1 << k;
    if (count == 0 || first + count > n) { set_error("zkhip_kzg_setup_range: [%zu, %zu) is not inside [0, 2^%u)", first, first + count, k); return ZKHIP_EINVAL; }
    el2<Fr> s = from_abi<Fr>(mem_load(s_u));
    void *d_sc, *d_pts;
    ZK_TRY(ctx->get_scratch("kzg_scalars", count * 32, &d_sc));
    ZK_TRY(ctx->get_scratch("kzg_points", count * 64, &d_pts));
    const el2<Fr> sn = pow_u64<Fr>(s, (uint64_t)n);
    const dim3 grid(div_up(div_up(count, 16), 64)), block(64);
    if (g) {
        hipLaunchKernelGGL(k_setup_scalars, grid, block, 0, ctx->stream, (uint32_t*)d_sc, count, first, s.v, s.v, s.v, 0);
        ZK_LAUNCH_CHECK();
        ZK_TRY(zkhip_fixed_base_mul_device(ctx, d_sc, count, d_pts));
        ZK_TRY(srs_build_raw(ctx, d_pts, count, g));
        srs_set_range(*g, first, n);
    }
    if (g_lagrange) {
        // l_i(s) = (s^n - 1)/n * w^i / (s - w^i), batch-inverted on the device; s - w^i = 0 for some i iff s^n = 1
        el2<Fr> omega = from_canonical_words<Fr>(FR_ROOT_OF_UNITY);
        for (uint32_t i = k; i < FR_S; ++i) omega = sqr(omega);
        const el2<Fr> vanish = reduce(sn - one<Fr>());
        if (is_zero(vanish)) { set_error("zkhip_kzg_setup: s is an n-th root of unity"); return ZKHIP_EINVAL; }
        const el2<Fr> num = vanish * inv<Fr>(from_u64<Fr>((uint64_t)n));
        hipLaunchKernelGGL(k_setup_scalars, grid, block, 0, ctx->stream, (uint32_t*)d_sc, count, first, omega.v, s.v, num.v, 1);
        ZK_LAUNCH_CHECK();
        ZK_TRY(zkhip_batch_invert_device(ctx, d_sc, count));
        hipLaunchKernelGGL(k_setup_scalars, grid, block, 0, ctx->stream, (uint32_t*)d_sc, count, first, omega.v, s.v, num.v, 2);
        ZK_LAUNCH_CHECK();
        ZK_TRY(zkhip_fixed_base_mul_device(ctx, d_sc, count, d_pts));
        ZK_TRY(srs_build_raw(ctx, d_pts, count, g_lagrange));
        srs_set_range(*g_lagrange, first, n);
    }
    return ZKHIP_OK;
}

extern "C" int zkhip_kzg_setup(zkhip_ctx* ctx, uint32_t k, const uint64_t s_u[4], zkhip_srs** g, zkhip_srs** g_lagrange) {
    if (k > 24) { set_error("zkhip_kzg_setup: k = %u unsupported (max 24)", k); return ZKHIP_EINVAL; }
    return zkhip_kzg_setup_range(ctx, k, s_u, 0, (size_t)1 << k, g, g_lagrange);
}
