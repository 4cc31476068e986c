import cProfile, pstats, os, sys, time
sys.path.insert(0, "/root/repo")
import halo2_zkcert_amd.ffi as ffi
import halo2_zkcert_amd.prover as pv
ctx = ffi.Context(0)
p = pv.Prover(pv.GpuBackend(ctx, ffi), pv.CircuitShape.rsa(17), satisfiable=True)
w = p.witness(0)
for _ in range(5): p.prove_native(w)
t0=time.perf_counter()
for _ in range(20): p.prove_native(w)
print("per proof ms", (time.perf_counter()-t0)/20*1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(20): p.prove_native(w)
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(12)
