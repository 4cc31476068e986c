"""MSM-only loop for rocprofv3 (python3 tools/msm_only.py [k] [ncols]): 4 batched commitments of ncols columns."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import halo2_zkcert_amd.ffi as ffi
import halo2_zkcert_amd.prover as pv

ctx = ffi.Context(0)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 22
ncols = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n = 1 << k
params = ffi.ParamsKZG.setup(ctx, k, pv.fr_from_int_host(0x1234567))
cols = [ctx.synth_fill(n, 100 + j) for j in range(ncols)]
ctx.to_host(params.commit_batch_device(cols))
ctx.profile_enable(True)
t0 = time.perf_counter()
for _ in range(4):
    ctx.to_host(params.commit_batch_device(cols))
dt = (time.perf_counter() - t0) / 4
print(f"k={k} ncols={ncols}: {dt * 1e3:.3f} ms per batch; digits {ctx.profile_read('msm_digits')[0] / 4:.3f} ms")
