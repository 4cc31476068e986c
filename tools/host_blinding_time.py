"""The one-call proof as a Rust caller would drive it: advice columns AND the blinding draws (incl. the vanishing argument's random polynomial, n scalars) handed over as
PAGEABLE host arrays (upstream draws them from the caller's rng: INTEGRATION.md 6).  Times prove_native at k = 22 / 19 for device inputs, host advice, host advice + host blinding."""
import os, sys, time, statistics
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "oracle"), os.path.join(os.getcwd(), "tests")]
import numpy as np, torch
import halo2_zkcert_amd.ffi as ffi
import halo2_zkcert_amd.prover as pv
ctx = ffi.Context(0)
shapes = (("agg22", pv.CircuitShape.agg(22, 3, 1), "evm"), ("sha19", pv.CircuitShape.sha256(19, n_advice=32, n_fixed=12), "poseidon"), ("rsa17", pv.CircuitShape.rsa(17), "poseidon"))
for name, sh, kind in [s_ for s_ in shapes if len(sys.argv) < 2 or s_[0] in sys.argv[1:]]:
    gp = pv.Prover(pv.GpuBackend(ctx, ffi), sh, satisfiable=True)
    w = gp.witness(0)
    bf, n = sh.blinding_factors, 1 << sh.k
    host = dict(lookup_permuted=ctx.to_host(ctx.synth_fill(max(1, 2 * len(sh.lookups)) * (bf + 1), 11)).copy(), perm_z=ctx.to_host(ctx.synth_fill(sh.n_perm_sets * bf, 12)).copy(),
                lookup_z=ctx.to_host(ctx.synth_fill(max(1, len(sh.lookups)) * bf, 13)).copy(), random_poly=ctx.to_host(ctx.synth_fill(n, 14)).copy())
    def med(f, reps=(25 if sh.k <= 17 else 6)):
        f(); torch.cuda.synchronize(); ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        return round(statistics.median(ts), 3)
    a = med(lambda: gp.prove_native(w, transcript=kind))
    b = med(lambda: gp.prove_native(w, transcript=kind, host_inputs="pageable"))
    c = med(lambda: gp.prove_native(w, transcript=kind, host_inputs="pageable", blinding=host))
    p_ = med(lambda: gp.prove_native(w, transcript=kind, host_inputs=True))
    print(f"{name}: device inputs {a} ms; pinned host advice {p_} ms; pageable host advice {b} ms; + pageable host blinding {c} ms", flush=True)
    gp.release(); gp.b.params.free(); del gp, w
    torch.cuda.empty_cache()
