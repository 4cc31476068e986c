"""cold vs warm host memory: a FRESH pageable array per call (what a Rust caller's Vec<Fr> is) against the same array again and again"""
import os, sys, time, statistics
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "oracle"), os.path.join(os.getcwd(), "tests")]
import numpy as np, torch
import halo2_zkcert_amd.ffi as ffi
import halo2_zkcert_amd.prover as pv
import zkoracle_py as zo
ctx = ffi.Context(0)
k = 22; n = 1 << k
p = ffi.ParamsKZG.setup(ctx, k, zo.fr_from_int(0x5EED0000 + k))
d = ctx.synth_fill(n, 77)
base = ctx.to_host(d).copy()
def timed(f, args_list):
    ts = []
    for a in args_list:
        t0 = time.perf_counter(); f(a); ts.append(round((time.perf_counter() - t0) * 1e3, 2))
    return ts
for reg in (1, 0):
    os.environ["X"] = "1"
    ctx.set_option("msm_host_chunks", 0 if reg else 1)
    fresh = [base.copy() for _ in range(8)]
    print(f"zkhip_msm_g1 2^22, {'pipelined (registers the slice)' if reg else 'one piece (no registration)'}: fresh array per call", timed(lambda a: p.commit(a), fresh), "same array", timed(lambda a: p.commit(a), [fresh[0]] * 8), flush=True)
p.free()
sh, kind = pv.CircuitShape.sha256(19, n_advice=32, n_fixed=12), "poseidon"
gp = pv.Prover(pv.GpuBackend(ctx, ffi), sh, satisfiable=True)
w = gp.witness(0)
gp.prove_native(w, transcript=kind, host_inputs="pageable")
def fresh_proof(_):
    w.pop("advice_host_pageable", None)
    cols = [c_.cpu() for c_ in w["advice"]]      # fresh pageable tensors
    w["advice_host_pageable"] = cols
    torch.cuda.synchronize()
    t0 = time.perf_counter(); gp.prove_native(w, transcript=kind, host_inputs="pageable"); torch.cuda.synchronize()
    return round((time.perf_counter() - t0) * 1e3, 1)
for thr, reg in ((1, 0), (0, 0), (0, 1), (1, 1)):
    ctx.set_option("host_copy_thread", thr)
    ctx.set_option("host_register", reg)
    print(f"sha19 proof, FRESH pageable advice per proof, host_copy_thread={thr} host_register={reg}:", [fresh_proof(i) for i in range(8)], flush=True)
ctx.set_option("host_copy_thread", 1); ctx.set_option("host_register", 0)
gp.release(); gp.b.params.free(); del gp, w; torch.cuda.empty_cache()
sh, kind = pv.CircuitShape.agg(22, 3, 1), "evm"
gp = pv.Prover(pv.GpuBackend(ctx, ffi), sh, satisfiable=True)
w = gp.witness(0)
bf, n = sh.blinding_factors, 1 << sh.k
host = dict(lookup_permuted=ctx.to_host(ctx.synth_fill(2 * (bf + 1), 11)).copy(), perm_z=ctx.to_host(ctx.synth_fill(sh.n_perm_sets * bf, 12)).copy(),
            lookup_z=ctx.to_host(ctx.synth_fill(bf, 13)).copy(), random_poly=ctx.to_host(ctx.synth_fill(n, 14)).copy())
gp.prove_native(w, transcript=kind, host_inputs="pageable")
def fresh22(blind):
    w.pop("advice_host_pageable", None)
    w["advice_host_pageable"] = [c_.cpu() for c_ in w["advice"]]
    b = dict(host, random_poly=host["random_poly"].copy()) if blind else None
    torch.cuda.synchronize()
    t0 = time.perf_counter(); gp.prove_native(w, transcript=kind, host_inputs="pageable", blinding=b); torch.cuda.synchronize()
    return round((time.perf_counter() - t0) * 1e3, 1)
for thr, reg in ((1, 0), (0, 0), (0, 1)):
    ctx.set_option("host_copy_thread", thr); ctx.set_option("host_register", reg)
    print(f"agg22 proof, FRESH pageable advice, thread={thr} register={reg}:", [fresh22(False) for _ in range(5)], "+ fresh host blinding:", [fresh22(True) for _ in range(5)], flush=True)
print("agg22 device inputs:", [round((lambda t0: (gp.prove_native(w, transcript=kind), torch.cuda.synchronize(), (time.perf_counter() - t0) * 1e3)[2])(time.perf_counter()), 1) for _ in range(4)])
