#!/usr/bin/env python3
"""Where does the FIRST proof of a fresh process go?  (a reference command is one proof per process,
/root/reference/src/bin/cli.rs:296-321,464-527 — `first_proof_s` is what a patched command sees.)

    python tools/cold_start.py [agg22|rsa17|sha19] [--tiny-first]

One process, wall clock per stage, with the library's own counters beside each stage: time inside hipMalloc (alloc_us), bytes allocated,
and — with --tiny-first — one k = 8 proof before anything large, which loads every code object of libzkhip.so and pays every
first-launch cost of the runtime without any data behind it: what remains in the k = 22 stages afterwards is allocation and real work.
Run it as the FIRST GPU process of a gpurun call for the fresh-box figure, and a second time in the same call for the warm-box one."""
import os
import sys
import time

t_proc = time.perf_counter()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
t0 = time.perf_counter()
import torch

t_torch = time.perf_counter() - t0
t0 = time.perf_counter()
import halo2_zkcert_amd.ffi as ffi
import halo2_zkcert_amd.prover as pv

ffi.lib()
t_lib = time.perf_counter() - t0

then = sys.argv[sys.argv.index("--then") + 1] if "--then" in sys.argv else None      # a second configuration in the same process (what bench.py's other configs see)
args = [a for a in sys.argv[1:] if not a.startswith("--") and a != then]
name = args[0] if args else "agg22"
tiny_first = "--tiny-first" in sys.argv
SHAPES = {"agg22": lambda: pv.CircuitShape.agg(22, 3, 1), "rsa17": lambda: pv.CircuitShape.rsa(17),
          "sha19": lambda: pv.CircuitShape.sha256(19, n_advice=32, n_fixed=12)}
KINDS = {"agg22": "evm", "rsa17": "poseidon", "sha19": "poseidon"}
shape, kind = SHAPES[name](), KINDS[name]
print(f"  {'import torch':46s} {t_torch:8.3f} s\n  {'import ffi + dlopen libzkhip.so':46s} {t_lib:8.3f} s", flush=True)

ctx = None
total = 0.0


def counters():
    if ctx is None:
        return (0, 0, 0)
    return tuple(ctx.profile_counter(n_) for n_ in ("alloc_us", "alloc_bytes", "alloc_calls"))


def clock(label, fn, count=True):
    global total
    c0 = counters()
    m0 = torch.cuda.memory_stats().get("reserved_bytes.all.current", 0) if ctx is not None else 0
    t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    c1 = counters()
    m1 = torch.cuda.memory_stats().get("reserved_bytes.all.current", 0)
    if count:
        total += dt
    print(f"  {label:46s} {dt:8.3f} s   hipMalloc inside the library: {(c1[0] - c0[0]) / 1e3:7.1f} ms for {(c1[1] - c0[1]) / 2**30:6.2f} GiB in "
          f"{c1[2] - c0[2]:3d} calls; torch allocator grew {(m1 - m0) / 2**30:6.2f} GiB", flush=True)
    return r


def make_ctx():
    global ctx
    ctx = ffi.Context(0)
    return ctx


clock("zkhip_init (hipInit, streams, pinned buffers)", make_ctx)
be = pv.GpuBackend(ctx, ffi)
if tiny_first:
    def tiny():
        p = pv.Prover(pv.GpuBackend(ctx, ffi), pv.CircuitShape.small(8), satisfiable=True)
        w = p.witness(0)
        p.prove_native(w, transcript=kind)
        p.release()
        p.b.params.free()
    clock("k = 8 proof (every code object, first launches)", tiny)
    clock("k = 8 proof again", tiny)
t_setup0 = time.perf_counter()
prover = clock("Prover: ParamsKZG.setup + keygen-shaped fixture", lambda: pv.Prover(be, shape, satisfiable=True))
wit = clock("witness fixture", lambda: prover.witness(0))
ctx.set_option("host_timing", int(os.environ.get("COLD_TIMING", "1")))
clock("first proof", lambda: prover.prove_native(wit, transcript=kind))
first_proof_s = time.perf_counter() - t_setup0
ctx.set_option("host_timing", 0)
clock("second proof", lambda: prover.prove_native(wit, transcript=kind), count=False)
if os.environ.get("COLD_TIMING_STEADY"):
    ctx.set_option("host_timing", 1)      # the host-side phase marks of a steady-state proof
clock("third proof", lambda: prover.prove_native(wit, transcript=kind), count=False)
ctx.set_option("host_timing", 0)
print(f"  first_proof_s as bench.py defines it (Prover + witness + first proof): {first_proof_s:.3f} s; process so far {time.perf_counter() - t_proc:.3f} s")

if then:
    print(f"  -- then {then} in the same process (release, empty_cache, new Prover): what bench.py's configs.{then}.first_proof_s is made of")
    clock("release the first key, free its params", lambda: (prover.release(), be.params.free()), count=False)
    del prover, wit
    clock("torch.cuda.empty_cache()", lambda: torch.cuda.empty_cache(), count=False)
    t1 = time.perf_counter()
    be2 = pv.GpuBackend(ctx, ffi)
    p2 = clock(f"Prover({then})", lambda: pv.Prover(be2, SHAPES[then](), satisfiable=True), count=False)
    w2 = clock("witness", lambda: p2.witness(0), count=False)
    clock("first proof", lambda: p2.prove_native(w2, transcript=KINDS[then]), count=False)
    print(f"  first_proof_s of {then} after {name}: {time.perf_counter() - t1:.3f} s")
    clock("second proof", lambda: p2.prove_native(w2, transcript=KINDS[then]), count=False)
