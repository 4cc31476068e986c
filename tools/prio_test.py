import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import halo2_zkcert_amd.ffi as ffi, halo2_zkcert_amd.prover as pv
mode = sys.argv[1] if len(sys.argv) > 1 else "default"
if mode == "high":
    hp = torch.cuda.Stream(priority=-1)
    torch.cuda.set_stream(hp)
ctx = ffi.Context(0)
ctx.use_torch_stream()
p = pv.Prover(pv.GpuBackend(ctx, ffi), pv.CircuitShape.rsa(17), satisfiable=True)
w = p.witness(0)
for _ in range(3): p.prove(w)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30): p.prove(w)
torch.cuda.synchronize()
print(mode, (time.perf_counter() - t0) / 30 * 1e3, "ms")
