import os, sys, time, statistics
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "oracle")]
import numpy as np
import halo2_zkcert_amd.ffi as ffi
import zkoracle_py as zo
ctx = ffi.Context(0)
for k in (20, 22):
    n = 1 << k
    p = ffi.ParamsKZG.setup(ctx, k, zo.fr_from_int(0x5EED0000 + k))
    d = [ctx.synth_fill(n, 900 + j) for j in range(4)]
    host = [ctx.to_host(c).copy() for c in d]
    def med(f, reps=7):
        f(); ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
        return round(statistics.median(ts), 3)
    def dev():
        p.commit_batch_device(d); ctx.synchronize()
    print(f"k={k}: 4 columns device-resident batch {med(dev)} ms; 4 x zkhip_msm_g1 {med(lambda: [p.commit(h) for h in host])} ms; zkhip_msm_g1_batch {med(lambda: p.commit_batch_host(host))} ms", flush=True)
    p.free()
