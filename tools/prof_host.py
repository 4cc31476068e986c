import sys, time, cProfile, pstats
sys.path.insert(0, '/root/repo')
import halo2_zkcert_amd.ffi as ffi, halo2_zkcert_amd.prover as pv
ctx = ffi.Context(0)
p = pv.Prover(pv.GpuBackend(ctx, ffi), pv.CircuitShape.rsa(17), satisfiable=True)
w = p.witness(0)
for _ in range(3): p.prove(w)
pr = cProfile.Profile(); pr.enable()
for _ in range(5): p.prove(w)
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
