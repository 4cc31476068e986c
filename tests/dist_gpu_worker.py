"""One rank of a multi-process proof on the GPU (TEST INFRASTRUCTURE; launched by tests/test_gpu_distributed.py through
torch.distributed.run).  Every rank joins the library's communicator (host-staged transport over gloo when the ranks share one
device — RCCL refuses that — or RCCL when each rank has its own GPU), builds the same satisfiable instance over its shard of the
SRS, proves with zkhip_create_proof_ex and writes the proof bytes; the test compares them with the single-GPU proof."""
import json
import os
import sys

ROOT = os.environ["ZK_ROOT"]
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]

import torch
import torch.distributed as dist

import halo2_zkcert_amd.ffi as ffi
import halo2_zkcert_amd.prover as pv

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
one_device = os.environ.get("ZK_ONE_DEVICE", "1") == "1"
dev = 0 if one_device else int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(dev)
if one_device:
    dist.init_process_group("gloo")
else:
    dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
ctx = ffi.Context(dev)
ctx.comm_init(rank, world, dist)
if os.environ.get("ZK_STALL_TEST") == "1":
    # the library's wait deadline (common.hpp wait_poll): one all-gather through the communicator, then a host wait on the context's stream.
    # The stand-in library stalls ONE rank's collective stream (ZKFAKE_RCCL_STALL=device:<rank>:<nth call>, set HERE: the stand-in counts calls
    # only while a plan is set, so the count does not depend on how many exchanges zkhip_comm_init's self-checks issued): that rank's wait must
    # fail with the deadline's message, the other rank's must succeed.  No teardown: a communicator with a stuck stream is not destroyed.
    os.environ["ZKFAKE_RCCL_STALL"] = "device:0:1"
    send = torch.full((64,), rank + 1, dtype=torch.uint8, device="cuda")
    recv = torch.zeros((64 * world,), dtype=torch.uint8, device="cuda")
    res = {"rank": rank, "error": None}
    torch.cuda.synchronize()
    try:
        ctx.comm_allgather(send, recv)
        ctx.synchronize()
        res["recv"] = recv.cpu().tolist()[::64]
    except Exception as e:   # noqa: BLE001
        res["error"] = str(e)
    if res["error"]:
        # what a Rust Drop / Python close() does after the deadline error must RETURN (zkhip_destroy used to hipStreamSynchronize the
        # stuck stream without a deadline).  The context stays dead across zkhip_comm_destroy (which resets the communicator's own stuck flag):
        # a later wait still fails at once; then close() abandons the device-side resources and comes back.
        import time

        t0 = time.time()
        res["ctx_dead"] = ctx.profile_counter("ctx_dead")
        ctx.comm_destroy()
        res["ctx_dead_after_comm_destroy"] = ctx.profile_counter("ctx_dead")
        try:
            ctx.synchronize()
            res["wait_after_destroy"] = "returned"
        except Exception as e:   # noqa: BLE001
            res["wait_after_destroy"] = str(e)
        ctx.close()
        res["close_s"] = time.time() - t0
    with open(os.path.join(os.environ["ZK_OUT"], f"rank{rank}.json"), "w") as f:
        json.dump(res, f)
    sys.stdout.flush()
    os._exit(0)
if os.environ.get("ZK_STALL_TEST") == "proof":
    # the same deadline met INSIDE a proof whose advice columns are the caller's host arrays (the error exit of zkhip_create_proof_ex
    # used to wait for the stuck stream without a deadline).  Rank 0's collective stream is stalled after the proof's LAST exchange (the partial
    # sums of SHPLONK's second commitment: the next thing the library does is a host wait for that commitment — the stand-in's own collectives
    # are synchronous, so a stall before an earlier exchange would be absorbed by the next exchange's hipStreamSynchronize inside the stand-in
    # instead of meeting the library's deadline).  Rank 0's proof must come back with the deadline's error well before the stall ends; rank 1's
    # proof completes (the exchange itself was served).  A watchdog thread ends a rank that waits for a peer that has left.
    import threading
    import time

    def leave(res):
        with open(os.path.join(os.environ["ZK_OUT"], f"rank{rank}.json"), "w") as f:
            json.dump(res, f)
        sys.stdout.flush()
        os._exit(0)
    threading.Timer(120.0, lambda: leave({"rank": rank, "error": "watchdog: still waiting after 120 s", "elapsed_s": 120.0})).start()
    ctx.comm_shard("points")
    p = pv.Prover(pv.GpuBackend(ctx, ffi), pv.CircuitShape.small(8), satisfiable=True)
    w = p.witness(1)
    p.prove_native(w, transcript="poseidon", host_inputs=True)      # the first proof of a key may issue one-time exchanges (the key's coset forms)
    c1 = ctx.comm_describe()["collectives"]
    p.prove_native(w, transcript="poseidon", host_inputs=True)      # a steady-state proof without a fault: how many exchanges a proof issues
    c2 = ctx.comm_describe()["collectives"]
    # the stand-in reads its fault plan at every call and counts calls only while a plan is set: from here on.  Stall rank 0's stream after
    # the LAST exchange of the next proof
    os.environ["ZKFAKE_RCCL_STALL"] = f"device:0:{c2 - c1}"
    res_plan = {"exchanges_per_proof": c2 - c1, "stall_at_call": c2 - c1}
    t0 = time.time()
    res = {"rank": rank, "error": None}
    try:
        p.prove_native(w, transcript="poseidon", host_inputs=True)
    except Exception as e:   # noqa: BLE001
        res["error"] = str(e)
    res["elapsed_s"] = time.time() - t0
    res.update(res_plan, exchanges_issued=ctx.comm_describe()["collectives"] - c2)
    if res["error"]:      # the same exit a caller takes: drop the context.  Must return
        t2 = time.time()
        try:                  # ... and a caller that tries once more is refused at once: nothing is queued behind the stuck streams, no upload thread is started
            p.prove_native(w, transcript="poseidon", host_inputs=True)
            res["second_call"] = "returned a proof"
        except Exception as e:   # noqa: BLE001
            res["second_call"] = str(e)
        res["second_call_s"] = time.time() - t2
        t1 = time.time()
        ctx.close()
        res["close_s"] = time.time() - t1
    leave(res)
mode = os.environ.get("ZK_SHARD_MODE", "points")
ctx.comm_shard(mode)
out = {"transport": ctx.transport, "shard_mode": mode}
native_only = os.environ.get("ZK_NATIVE_ONLY") == "1"       # full-size cases: the Python schedule (extended domain, no workspace reuse) is not run
for spec in json.loads(os.environ["ZK_SHAPES"]):
    if spec[0] == "small":
        sh = pv.CircuitShape.small(spec[1])
    elif spec[0] == "sha":
        sh = pv.CircuitShape.sha256(spec[1], n_advice=12, n_fixed=5)
    elif spec[0] == "phase":                                 # an advice column of the second phase + a user challenge
        sh = pv.CircuitShape.two_phase(spec[1])
    elif spec[0] == "two":                                   # two lookups (agg-like with two lookup-advice columns): owner mapping of a', s', z per lookup
        sh = pv.CircuitShape(f"two_lookups_k{spec[1]}", spec[1], 2, 2, 1, 4, 6, 0x2100C0 + spec[1])
    elif spec[0] == "shafull":                               # bench.py's SHA-256-shaped configuration (BASELINE configs[2])
        sh = pv.CircuitShape.sha256(spec[1], n_advice=32, n_fixed=12)
    elif spec[0] == "agg":                                   # bench.py's aggregation-shaped configuration (BASELINE configs[3])
        sh = pv.CircuitShape.agg(spec[1], 3, 1)
    else:
        sh = pv.CircuitShape.rsa(spec[1])
    p = pv.Prover(pv.GpuBackend(ctx, ffi), sh, satisfiable=True)
    first, count, total = p.b.params.range()
    assert total == 1 << sh.k and (first, count) == (ctx.shard_range(total) if mode == "points" else (0, total)), (first, count, total)
    w = p.witness(int(os.environ.get("ZK_WITNESS", "1")))
    g0 = ctx.comm_bytes_gathered()
    tr = p.prove_native(w, transcript=spec[2])
    gathered = ctx.comm_bytes_gathered() - g0
    # the Python schedule over the small entry points: MSMs collective, the rest replicated
    tp = tr if native_only else p.prove(w, transcript=spec[2])
    out[f"{spec[0]}{spec[1]}{spec[2]}"] = dict(native=tr["proof"].hex(), python=tp["proof"].hex(), bytes_gathered=gathered)
    p.release()
    p.b.params.free()
    del p, w, tr, tp
    ctx.trim()
    torch.cuda.empty_cache()
out["bytes_gathered"] = ctx.comm_bytes_gathered()
out["comm"] = ctx.comm_describe()
out["modes"] = {k_: ctx.profile_counter(k_) for k_ in ("proofs_row_sharded", "proofs_pieces_sharded", "shplonk_row_sharded")}
with open(os.path.join(os.environ["ZK_OUT"], f"rank{rank}.json"), "w") as f:
    json.dump(out, f)
dist.barrier()
ctx.comm_destroy()
dist.destroy_process_group()
