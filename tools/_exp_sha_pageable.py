import os, sys, time, statistics
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "oracle"), os.path.join(os.getcwd(), "tests")]
import numpy as np, torch
import halo2_zkcert_amd.ffi as ffi
import halo2_zkcert_amd.prover as pv
ctx = ffi.Context(0)
def med(f, reps=6):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return round(statistics.median(ts), 3)
mode = sys.argv[1]
# history: an aggregation-shaped k = 22 prover first
sh22 = pv.CircuitShape.agg(22, 3, 1)
g22 = pv.Prover(pv.GpuBackend(ctx, ffi), sh22, satisfiable=True)
w22 = g22.witness(0)
bf, n = sh22.blinding_factors, 1 << 22
host22 = dict(lookup_permuted=ctx.to_host(ctx.synth_fill(2 * (bf + 1), 11)).copy(), perm_z=ctx.to_host(ctx.synth_fill(sh22.n_perm_sets * bf, 12)).copy(),
              lookup_z=ctx.to_host(ctx.synth_fill(bf, 13)).copy(), random_poly=ctx.to_host(ctx.synth_fill(n, 14)).copy())
if mode == "dev":
    print("k22 device", med(lambda: g22.prove_native(w22, transcript="evm")))
elif mode == "pageable":
    print("k22 pageable", med(lambda: g22.prove_native(w22, transcript="evm", host_inputs="pageable")))
elif mode == "blinding":
    print("k22 pageable+blinding", med(lambda: g22.prove_native(w22, transcript="evm", host_inputs="pageable", blinding=host22)))
if len(sys.argv) > 2:
    g22.release(); g22.b.params.free(); del g22, w22; torch.cuda.empty_cache()
sh, kind = pv.CircuitShape.sha256(19, n_advice=32, n_fixed=12), "poseidon"
gp = pv.Prover(pv.GpuBackend(ctx, ffi), sh, satisfiable=True)
w = gp.witness(0)
ts = []
for _ in range(24):
    t0 = time.perf_counter(); gp.prove_native(w, transcript=kind, host_inputs="pageable"); torch.cuda.synchronize(); ts.append(round((time.perf_counter() - t0) * 1e3, 1))
print("sha19 pageable, 24 proofs in a row:", ts, flush=True)
print("sha19 pinned:", med(lambda: gp.prove_native(w, transcript=kind, host_inputs=True)), flush=True)
print("sha19 pageable again:", med(lambda: gp.prove_native(w, transcript=kind, host_inputs="pageable")), flush=True)
if len(sys.argv) > 3:
    pass
