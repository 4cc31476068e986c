"""Host-side profile of Prover.prove (where the Python between the GPU launches goes)."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import halo2_zkcert_amd.ffi as ffi
import halo2_zkcert_amd.prover as pv

ctx = ffi.Context(0)
p = pv.Prover(pv.GpuBackend(ctx, ffi), pv.CircuitShape.rsa(17), satisfiable=True)
w = p.witness(0)
for _ in range(3):
    p.prove(w)
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    p.prove(w)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(14)
st.sort_stats("cumulative").print_stats(45)
