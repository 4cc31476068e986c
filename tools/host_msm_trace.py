"""for rocprofv3 --kernel-trace --stats: 12 pipelined zkhip_msm_g1 calls on a 2^22 pageable slice (the new k_merge_buckets among the MSM's kernels)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import halo2_zkcert_amd.ffi as ffi
import zkoracle_py as zo
ctx = ffi.Context(0)
k = 22
p = ffi.ParamsKZG.setup(ctx, k, zo.fr_from_int(0x5EED0000 + k))
host = ctx.to_host(ctx.synth_fill(1 << k, 77)).copy()
for _ in range(12):
    p.commit(host)
