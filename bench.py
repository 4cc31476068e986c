#!/usr/bin/env python3
"""bench.py — create_proof-shaped pass over the HIP hot path (MSM + NTT + quotient sweep).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one pass of halo2_zkcert_amd.prover.Prover.prove over the RSA k=17 synthetic table
(BASELINE.json configs[1]): 16 MSM_2^17, 11 iNTT_2^17, 11 NTT_2^19 + 1 iNTT_2^19, one 2^19-row sweep, the lookup
theta-compression, the permutation / lookup grand products and the evaluations at x, with a host round trip at
every Fiat-Shamir point.  Inputs (witness columns, SRS, pk cosets) are resident
in HBM before the timed region.  Prints ONE JSON line (rank 0).

N > 1: one process per GPU over RCCL.  Default: the path partitions by proof (the reference's leaf proofs are
independent, SURVEY.md §3.5 / BASELINE config 5) — every rank runs its own pass on its own witness, no
data-path collective, "scaling": "weak"; `value` is the wall time of the N-proof job (max over ranks).
--shard-msm instead splits ONE proof: every MSM is point-range sharded over the ranks, the 96-byte partial sums
are all-gathered and folded, NTTs and the sweep are replicated -> "strong" (DESIGN.md §multi-GPU).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def host_threads():
    """Cores this process may actually use: scheduler affinity, capped by a cgroup CPU quota if one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = max(1, min(n, q // per))
        except Exception:
            pass
    return n


def cpu_baseline(shape, threads):
    """The CPU oracle (oracle/zkoracle.c, OpenMP) running the same schedule once on the host cores.
    kind = "port": the reference's rayon prover cannot be built here (no Rust; un-vendored crates)."""
    sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
    import halo2_zkcert_amd.prover as pv
    from oracle_backend import OracleBackend

    p = pv.Prover(OracleBackend(threads), shape, satisfiable=shape.name.startswith("rsa"))
    w = p.witness(0)
    t0 = time.perf_counter()
    p.prove(w)
    dt = time.perf_counter() - t0
    return dict(value=round(dt, 4), unit="s", cores=threads, kind="port",
                sample=f"1 full pass of the same schedule ({shape.name}), setup excluded")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--k", type=int, default=17)
    ap.add_argument("--shape", default="rsa", choices=["rsa", "sha256"], help="circuit shape (BASELINE configs[1] / configs[2])")
    ap.add_argument("--witness", default="uniform", choices=["uniform", "survey"],
                    help="sha256 shape only: uniform field elements (worst case) or SURVEY.md 8(d)'s mix of 90 %% bits / 10 %% 32-bit words")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--shard-msm", action="store_true", help="N > 1: split one proof (strong scaling) instead of one proof per GPU")
    ap.add_argument("--shard-ntt", action="store_true", help="with --shard-msm: also distribute the coset NTTs by polynomial (all-gather)")
    ap.add_argument("--shard-sweep", action="store_true", help="with --shard-msm: also evaluate the quotient sweep by row range (all-gather of h)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short extra runs of the metric's other two configurations (SHA256-shaped k=19, aggregation-shaped k=22)")
    ap.add_argument("--python-schedule", action="store_true",
                    help="drive the proof from prover.py over the small entry points instead of zkhip_create_proof (same proof)")
    args = ap.parse_args()

    import torch

    import halo2_zkcert_amd.ffi as ffi
    import halo2_zkcert_amd.prover as pv

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist

        # test hooks (a 1-GPU box can still run the N > 1 control flow): ZKHIP_BENCH_ONE_DEVICE=1 puts every rank on device 0,
        # ZKHIP_BENCH_DIST_BACKEND=gloo replaces RCCL, which refuses two ranks on one device
        if os.environ.get("ZKHIP_BENCH_ONE_DEVICE") == "1":
            local_rank = 0
        torch.cuda.set_device(local_rank)
        be = os.environ.get("ZKHIP_BENCH_DIST_BACKEND", "nccl")
        if be == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(be)
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)

    ctx = ffi.Context(local_rank)
    shape = pv.CircuitShape.rsa(args.k) if args.shape == "rsa" else pv.CircuitShape.sha256(args.k)
    backend = pv.GpuBackend(ctx, ffi)
    shard = world > 1 and args.shard_msm
    if shard:
        backend = pv.ShardedCommit(backend, rank, world, dist, shard_ntt=args.shard_ntt, shard_sweep=args.shard_sweep)
    prover = pv.Prover(backend, shape, satisfiable=args.shape == "rsa")   # rsa shape: a satisfiable instance, i.e. a valid proof
    wit = prover.witness(0 if (shard or world == 1) else rank, dist=args.witness)   # one independent proof per rank unless sharding one
    n = 1 << shape.k
    counts = shape.counts(prover.dom.extended_k)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # one step = one create_proof: the library's own schedule (zkhip_create_proof, transcript through callbacks) unless the proof
    # is sharded over ranks or --python-schedule asks for the Python one (identical proofs: tests/test_gpu_prover.py)
    native = not shard and not args.python_schedule
    prove = prover.prove_native if native else prover.prove
    for _ in range(args.warmup):
        prove(wit)
    # Live HIP-event timing inside the timed region covers the dominant kernel only (every recorded span costs two event
    # records on the launch stream: ~60 spans are ~4 % of a 10 ms proof); the full per-kernel breakdown comes from extra,
    # untimed passes afterwards.
    DOMINANT = "msm_accum_affine"
    ctx.profile_select(DOMINANT)
    ctx.profile_enable(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        trace = prove(wit)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt * 1000.0 / args.steps
    ms, launches = ctx.profile_read(DOMINANT)
    dominant = dict(ms_per_step=round(ms / args.steps, 4), launches_per_step=launches / args.steps)

    kernels = {}
    extra = 3
    ctx.profile_select(None)
    ctx.profile_enable(True)
    for _ in range(extra):
        prove(wit)
    for name in ("msm_digits", "msm_plan", "msm_accum_affine", "msm_accum_jac", "msm_tail", "ntt_strided", "ntt_final", "sweep",
                 "lookup_permute", "grand_product", "eval_polynomial", "linear_combination", "kate_division"):
        ms, launches = ctx.profile_read(name)
        kernels[name] = dict(ms_per_step=round(ms / extra, 4), launches_per_step=launches / extra)
    ctx.profile_enable(False)
    barrier()

    if rank == 0:
        # dominant kernel: MSM bucket accumulation.  Algorithmic bytes = 96 B per (scalar, point) pair
        # (SURVEY.md §8(d)); one step issues `msm` columns of n/world pairs in 7 launches.
        acc = dominant   # measured inside the timed region
        pairs_per_step = counts["msm"] * (n // world if shard else n)
        alg_bytes_per_launch = 96.0 * pairs_per_step / max(acc["launches_per_step"], 1)
        avg_launch_s = acc["ms_per_step"] / max(acc["launches_per_step"], 1) / 1000.0
        achieved = alg_bytes_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        # the roofline that actually binds the kernel: the 32-bit integer multiplier (v_mad_u64_u32), measured at
        # 29.8 T lane-ops/s chip-wide (profiles/r01_microbench_gfx950.txt).  One pair costs W windows x one XYZZ mixed
        # addition = 8 products (171 mads) + 2 squarings (126 mads).
        c_bits, windows = prover.b.params.window()
        mads_per_step = pairs_per_step * windows * (8 * 171 + 2 * 126)
        int_achieved = mads_per_step / (acc["ms_per_step"] / 1000.0) / 1e12 if acc["ms_per_step"] > 0 else 0.0
        out = {
            "metric": "create_proof wall-time (s): RSA k=17 / SHA256 k=19 / agg k=22 at 1/2/4/8 GPU",
            "value": round(ms_per_step / 1000.0, 6), "unit": "s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": False, "scaling": "strong" if shard else "weak", "vs_baseline": None,
            "proofs_per_step": 1 if (shard or world == 1) else world,
            "dtype": "u256 (BN254 Fr/Fq, Montgomery, 9 x 29-bit limbs in registers / 8 x u32 in HBM)", "data": "synthetic",
            "config": {"workload": f"create_proof-shaped hot-path pass, {shape.name}: {counts['msm']} MSM_2^{shape.k} + "
                                   f"{counts['intt_n']} iNTT_2^{shape.k} + {counts['ntt_ext']} NTT_2^{prover.dom.extended_k} + "
                                   f"1 iNTT_2^{prover.dom.extended_k} + 1 sweep over 2^{prover.dom.extended_k} rows + lookup compression, "
                                   f"{shape.n_perm_sets}+{len(shape.lookups)} grand products, evaluations at x; "
                                   "lookup permute (sort) and SHPLONK multi-open computed; " + ("synthetic SATISFIABLE instance (gates, copy constraints, lookup hold: the output is a valid proof, see tests/test_gpu_prover.py::test_rsa_k17_valid_proof); " if args.shape == "rsa" else ("uniform synthetic witness; " if args.witness == "uniform" else "synthetic witness with SURVEY 8(d)'s value mix (90 % bits, 10 % words < 2^32); ")) +
                                   "halo2 Blake2bWrite transcript (restated, in the library; the reference's commands use Poseidon / Keccak)",
                       "k": shape.k, "advice": shape.n_advice, "fixed": shape.n_fixed, "lookups": len(shape.lookups),
                       "perm_columns": len(shape.perm_columns), "degree": shape.degree,
                       "host": "zkhip_create_proof (schedule in the library, transcript callbacks)" if native else "prover.py (Python schedule over the C ABI)",
                       "parallelism": "1 GPU" if world == 1 else (f"one proof, MSM point-range sharded x{world}, " + ("coset NTTs by polynomial + all-gather, " if args.shard_ntt else "NTTs replicated, ") + ("sweep by row range + all-gather" if args.shard_sweep else "sweep replicated") if shard
                                                                   else f"{world} independent proofs, one per GPU, no collective")},
            "roofline": {"kernel": "msm_accum_affine (k_accum_affine)", "bound": "hbm", "achieved": round(achieved, 2), "peak": 8000.0,
                         "unit": "GB/s", "frac": round(achieved / 8000.0, 5),
                         # HBM bytes per launch from the PMC passes committed in profiles/r01_v5_pmc_fetch_write.csv
                         # (FETCH_SIZE + WRITE_SIZE of k_accum_affine over 112 column-MSMs: 172.2 MB + 19.8 MB per 2^17 x 16-window
                         # column = 91.6 B per (pair, window)), scaled to this launch shape; not re-measured live.
                         "traffic": round(91.6 * (pairs_per_step / max(acc["launches_per_step"], 1)) * windows),
                         "algorithmic_bytes_per_launch": round(alg_bytes_per_launch),
                         "avg_launch_ms": round(avg_launch_s * 1000.0, 4),
                         "note": "MSM is integer-multiply bound, not HBM bound: see int_roofline and DESIGN.md"},
            "int_roofline": {"kernel": "msm_accum_affine", "bound": "v_mad_u64_u32 issue", "achieved": round(int_achieved, 2), "peak": 29.8,
                             "unit": "Tmad/s", "frac": round(int_achieved / 29.8, 4), "window_bits": c_bits, "windows": windows},
            "kernels_ms_per_step": kernels,
            **({"int_roofline_note": "sparse-digit witness: most (scalar, window) pairs are zero digits and are skipped, so the dense-pair "
                                      "mad count behind int_roofline / roofline.achieved does not describe this run"} if args.witness == "survey" else {}),
            "kernels_note": f"per-kernel HIP-event times from {extra} extra untimed passes; roofline/int_roofline use the dominant kernel's events recorded inside the timed region",
        }
        # The metric names three configurations; `value` is configs[1] (RSA k=17).  The other two are timed here with a few steps each
        # (same step = one zkhip_create_proof call, uniform synthetic witness for the SHA shape, satisfiable instance for k=22) so that the
        # line carries all three.  Never allowed to break the main measurement.
        if world == 1 and not args.no_other_configs and args.k == 17 and args.shape == "rsa":
            others = {}
            del prover, wit, trace
            for name, shp, sat, dist in (("sha256_shaped_k19", pv.CircuitShape.sha256(19), False, "uniform"),
                                         ("sha256_shaped_k19_bit_witness", pv.CircuitShape.sha256(19), False, "survey"),
                                         ("aggregation_shaped_k22", pv.CircuitShape.rsa(22), True, "uniform")):
                try:
                    torch.cuda.empty_cache()
                    p2 = pv.Prover(pv.GpuBackend(ctx, ffi), shp, satisfiable=sat)
                    w2 = p2.witness(0, dist=dist)
                    p2.prove_native(w2)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(3):
                        p2.prove_native(w2)
                    torch.cuda.synchronize()
                    others[name] = dict(value=round((time.perf_counter() - t0) / 3, 6), unit="s", steps=3, warmup=1)
                    p2.b.params.free()
                    del p2, w2
                except Exception as e:   # noqa: BLE001
                    others[name] = dict(error=str(e)[:200])
            out["other_configs"] = others
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(shape, host_threads())
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
