#!/usr/bin/env python3
"""Stress of the multi-rank proof path on ONE GPU (host-staged transport over gloo): every rank proves the same rotating instances
(shape, k, seed, transcript, shard mode) through the library's communicator AND, on a second context without a communicator, alone —
the two proofs must be the same bytes on every rank.  Launch under torch.distributed.run:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node=4 --master-addr 127.0.0.1 --master-port 29577 tools/stress_dist.py --seconds 300"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import torch
import torch.distributed as dist

import halo2_zkcert_amd.ffi as ffi
import halo2_zkcert_amd.prover as pv

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=300)
ap.add_argument("--kmin", type=int, default=8)
ap.add_argument("--kmax", type=int, default=12)
args = ap.parse_args()
torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
shapes = {"small": pv.CircuitShape.small, "agg": pv.CircuitShape.agg, "two": lambda k: pv.CircuitShape(f"two_lookups_k{k}", k, 2, 2, 1, 4, 6, 0x2100C0 + k),
          "sha": lambda k: pv.CircuitShape.sha256(k, n_advice=12, n_fixed=5), "phase": pv.CircuitShape.two_phase}
solo = ffi.Context(0)
done, seed, modes = 0, 0, {}
t_end = time.time() + args.seconds
stop = torch.zeros(1, dtype=torch.int32)
while True:
    stop[0] = 1 if time.time() > t_end else 0
    dist.broadcast(stop, src=0)          # every rank leaves the loop at the same iteration
    if int(stop[0]):
        break
    k = args.kmin + seed % (args.kmax - args.kmin + 1)
    name = ("small", "agg", "two", "sha", "phase")[(seed // 2) % 5]
    mode = ("points", "columns")[(seed // 5) % 2]
    kind = ("poseidon", "evm", "blake2b")[seed % 3]
    ctx = ffi.Context(0)
    ctx.comm_init(rank, world, dist)
    ctx.comm_shard(mode)
    if seed % 7 == 3:
        ctx.set_option("row_sharded", 0)
    ps = pv.Prover(pv.GpuBackend(ctx, ffi), shapes[name](k), satisfiable=True)
    pa = pv.Prover(pv.GpuBackend(solo, ffi), shapes[name](k), satisfiable=True)
    ws, wa = ps.witness(seed), pa.witness(seed)
    a = ps.prove_native(ws, transcript=kind, host_inputs=(seed % 4 == 0 and os.environ.get("STRESS_NO_HOST") != "1"))["proof"]
    b = pa.prove_native(wa, transcript=kind)["proof"]
    if a != b:
        diff_cols = [i for i, (x_, y_) in enumerate(zip(ws["advice"], wa["advice"])) if not (ctx.to_host(x_) == solo.to_host(y_)).all()]
        a2 = ps.prove_native(ws, transcript=kind)["proof"]
        b2 = pa.prove_native(wa, transcript=kind)["proof"]
        ws2 = ps.witness(seed)
        regen = [i for i, (x_, y_) in enumerate(zip(ws["advice"], ws2["advice"])) if not (ctx.to_host(x_) == ctx.to_host(y_)).all()]
        wa2 = pa.witness(seed)
        regen_solo = [i for i, (x_, y_) in enumerate(zip(wa["advice"], wa2["advice"])) if not (solo.to_host(x_) == solo.to_host(y_)).all()]
        print("MISMATCH", name, k, seed, mode, kind, rank, "witness columns that differ:", diff_cols, "re-proved: a same", a2 == a, "b same", b2 == b, "a2==b2", a2 == b2,
              "sharded-ctx witness regenerated differs in", regen, "solo witness regenerated differs in", regen_solo, flush=True)
        if os.environ.get("STRESS_CONTINUE") != "1":
            raise SystemExit(1)
    for c in ("proofs_row_sharded", "proofs_pieces_sharded", "shplonk_row_sharded"):
        modes[c] = modes.get(c, 0) + ctx.profile_counter(c)
    ps.release(); pa.release(); ps.b.params.free(); pa.b.params.free()
    del ps, pa, ws, wa
    dist.barrier()
    ctx.comm_destroy()
    ctx.close()
    done += 1
    seed += 1
if rank == 0:
    print(f"dist stress ok: {done} proofs x {world} ranks, sharded == single-GPU bytes; exchange modes taken {modes}")
dist.barrier()
dist.destroy_process_group()
