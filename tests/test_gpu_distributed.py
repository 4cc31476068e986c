"""GPU: one proof sharded over several ranks through the library's communicator (zkhip_comm_*).
A one-GPU box cannot give every rank its own device, and RCCL refuses two ranks per device, so the multi-rank cases run the library's
host-staged transport (the same sharding, folds and in-place all-gathers; only the exchange primitive differs) with both ranks on
device 0; the RCCL primitive itself is exercised with a one-rank communicator.  With >= 2 devices the same worker runs over RCCL."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import halo2_zkcert_amd.prover as pv

pytestmark = pytest.mark.gpu


def _free_port():
    """a TCP port that is free right now (the rendezvous of the multi-process tests)"""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# small 8 / 10: the coset-quotient path with row-sharded cosets (all-to-all of row windows) whenever 64 N divides n — N = 2 at k = 8, N = 8 at
# k = 10; N = 3 / 5 and the SHA shape (extended domain) take the all-gather path
SHAPES = [["small", 8, "poseidon"], ["sha", 9, "poseidon"], ["small", 7, "evm"], ["small", 10, "evm"], ["two", 9, "poseidon"],
          ["phase", 10, "poseidon"]]      # "phase": an advice column of the second phase and a user challenge (CircuitShape.two_phase)


_REF = {}


def _single_gpu_proofs(zk):
    if _REF:
        return _REF
    ffi, ctx = zk
    out = _REF
    for spec in SHAPES:
        sh = (pv.CircuitShape.small(spec[1]) if spec[0] == "small" else pv.CircuitShape.sha256(spec[1], n_advice=12, n_fixed=5) if spec[0] == "sha" else
              pv.CircuitShape.two_phase(spec[1]) if spec[0] == "phase" else
              pv.CircuitShape(f"two_lookups_k{spec[1]}", spec[1], 2, 2, 1, 4, 6, 0x2100C0 + spec[1]))
        p = pv.Prover(pv.GpuBackend(ctx, ffi), sh, satisfiable=True)
        out[f"{spec[0]}{spec[1]}{spec[2]}"] = p.prove_native(p.witness(1), transcript=spec[2])["proof"].hex()
        p.b.params.free()
    return out


def _run_workers(tmp_path, world, one_device, port, shapes=None, extra_env=None, timeout=900):
    env = dict(os.environ, ZK_ROOT=ROOT, ZK_OUT=str(tmp_path), ZK_SHAPES=json.dumps(shapes or SHAPES), ZK_ONE_DEVICE="1" if one_device else "0",
               HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_gpu_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return [json.load(open(tmp_path / f"rank{i}.json")) for i in range(world)]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_proof_equals_single_gpu_proof(zk, tmp_path, world):
    """N ranks (sharing device 0, host-staged transport): every rank holds 1/N of the SRS window tables, every MSM is a collective
    (slice sums, all-gather of the 96-byte partial sums, device fold), the coset NTTs go by polynomial and the sweep by row range
    (N = 2; with N = 3 the row count does not divide and the sweep stays replicated) — and every rank ends with exactly the bytes
    the single-GPU prover produces, under Poseidon and under Keccak, from zkhip_create_proof_ex and from the Python schedule."""
    ref = _single_gpu_proofs(zk)
    outs = _run_workers(tmp_path, world, True, 29611 + world)
    for o in outs:
        assert o["transport"] == "host"
        for key, hexs in ref.items():
            assert o[key]["native"] == hexs, (key, "native")
            assert o[key]["python"] == hexs, (key, "python")
        if world == 2:      # 64 N divides n for the k = 8 and k = 10 circuits: row windows, row-range pieces and SHPLONK on row ranges were taken
            assert o["modes"]["proofs_row_sharded"] >= 2 and o["modes"]["proofs_pieces_sharded"] >= 2 and o["modes"]["shplonk_row_sharded"] >= 2, o["modes"]
        else:               # N = 3: nothing divides — the all-gather path
            assert o["modes"]["proofs_row_sharded"] == 0, o["modes"]


def test_rccl_communicator_single_rank(zk):
    """The RCCL path of the library (dlopen'ed librccl, ncclCommInitRank, ncclAllGather on the communicator's stream fenced by
    events) with the one rank a one-GPU box allows: an all-gather is the identity, and a proof on a context that has a communicator
    is unchanged."""
    import torch

    ffi, _ = zk
    ctx = ffi.Context(0)
    try:
        ctx.comm_init(0, 1, None, transport="rccl")
        a = ctx.synth_fill(1000, 77)
        b = torch.zeros_like(a)
        ctx.comm_trace(True)          # the per-exchange trace (zkhip_comm_trace): one entry per exchange since the mark, outside a proof the phase is ""
        ctx.comm_allgather(a, b)
        ctx.synchronize()
        assert (ctx.to_host(a) == ctx.to_host(b)).all()
        ent, end_us = ctx.comm_trace_read()
        assert len(ent) == 1 and ent[0]["kind"] == "allgather" and ent[0]["phase"] == "" and ent[0]["bytes_received"] == 0 and not ent[0]["bulk"]
        assert 0 <= ent[0]["host_issue_us"] and 0 < ent[0]["stream_done_us"] <= end_us + 1.0
        assert ctx.comm_trace_read(cap=0)[0] == []      # a too-small caller buffer truncates, it does not overflow
        ctx.comm_trace(False)
        ctx.comm_allgather(a, b)
        ctx.comm_trace(True)          # restarting clears the record; nothing is recorded while it was off
        assert ctx.comm_trace_read()[0] == []
        ctx.comm_trace(False)
        sh = pv.CircuitShape.small(7)
        p = pv.Prover(pv.GpuBackend(ctx, ffi), sh, satisfiable=True)
        w = p.witness(0)
        t1 = p.prove_native(w, transcript="poseidon")["proof"]
        ctx.comm_destroy()
        assert p.prove_native(w, transcript="poseidon")["proof"] == t1
    finally:
        ctx.close()


def test_every_rccl_entry_point_meets_the_real_librccl_with_one_rank(zk):
    """VERDICT r5 item 2.  Everything the multi-GPU path calls in RCCL, against the REAL library, with the one rank this pool allows: option
    comm_selfcheck_force makes zkhip_comm_init run its self-checks on a one-rank communicator with the pair (rank, rank) in every group —
    a grouped ncclSend / ncclRecv to self on the communicator's stream, ncclCommSplit(comm, 0, rank, &bulk, NULL), the tagged exchange on the
    split on ITS stream, the verdict / readiness all-gathers, ncclCommCount / ncclCommUserRank on both — and zkhip_comm_destroy the split before
    its parent.  A k = 10 proof on that context is unchanged.  Left unexercised: the N > 1 transport itself."""
    import ctypes

    ffi, ctx0 = zk
    sh = pv.CircuitShape.small(10)
    p0 = pv.Prover(pv.GpuBackend(ctx0, ffi), sh, satisfiable=True)
    want = bytes(p0.prove_native(p0.witness(0), transcript="poseidon")["proof"])
    p0.release()
    p0.b.params.free()
    ctx = ffi.Context(0)
    try:
        ctx.set_option("comm_selfcheck_force", 1)
        for cycle in range(2):        # init -> use -> destroy twice on one context: the destroy order leaves nothing behind
            ctx.comm_init(0, 1, None, transport="rccl")
            assert ctx.profile_counter("comm_bulk") == 1
            assert ctx.profile_counter("comm_selfcheck") == 1 | 2 | 4 | 8 | 16      # a2a, split, a2a on the split, counts agree, self pairs (comm.hip SC_*)
            d = ctx.comm_describe()
            assert d["transport"] == "rccl" and d["transport_ranks"] == 1 and d["nranks"] == 1 and d["collectives"] == 0
            if cycle == 0:
                # the library that served those calls is the real one: the mapped librccl (torch's copy), no stand-in in this process
                maps = open("/proc/self/maps").read()
                assert "fake_rccl" not in maps and "replay_rccl" not in maps
                paths = sorted({ln.split()[-1] for ln in maps.splitlines() if "librccl" in ln})
                assert paths, "no librccl mapped"
                ver = ctypes.c_int(0)
                assert ctypes.CDLL(paths[0]).ncclGetVersion(ctypes.byref(ver)) == 0 and ver.value >= 20700, ver.value   # send / recv to self: 2.7+
                print(f"real librccl: {paths[0]} version code {ver.value}")
            p = pv.Prover(pv.GpuBackend(ctx, ffi), sh, satisfiable=True)
            assert bytes(p.prove_native(p.witness(0), transcript="poseidon")["proof"]) == want
            p.release()
            p.b.params.free()
            ctx.comm_destroy()
            assert ctx.profile_counter("comm_bulk") == 0 and ctx.profile_counter("comm_selfcheck") == 0
        # without the option a one-rank communicator runs no self-check and has no bulk companion (the default)
        ctx.set_option("comm_selfcheck_force", 0)
        ctx.comm_init(0, 1, None, transport="rccl")
        assert ctx.profile_counter("comm_bulk") == 0 and ctx.profile_counter("comm_selfcheck") == 0
        ctx.comm_destroy()
    finally:
        ctx.close()


def test_multi_device_rccl_if_available(zk, tmp_path):
    """with >= 2 GPUs: the same worker, one rank per device, RCCL transport"""
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    ref = _single_gpu_proofs(zk)
    outs = _run_workers(tmp_path, 2, False, 29631)
    for o in outs:
        assert o["transport"] == "rccl" and o["bytes_gathered"] > 0
        for key, hexs in ref.items():
            assert o[key]["native"] == hexs


@pytest.mark.parametrize("world,mode", [(8, "points"), (8, "columns"), (5, "points"), (4, "points")])
def test_eight_and_five_ranks(zk, tmp_path, world, mode):
    """the rank count of the target node (8) and an odd one (5), all sharing device 0 through the host-staged transport: more ranks
    than columns in a batch (padded all-gather rounds, ranks without a column), uneven point ranges, row ranges of 1/8 — every rank's
    proof bytes equal the single-GPU proof's"""
    ref = _single_gpu_proofs(zk)
    os.environ["ZK_SHARD_MODE"] = mode
    try:
        outs = _run_workers(tmp_path, world, True, 0)
    finally:
        os.environ.pop("ZK_SHARD_MODE", None)
    assert len(outs) == world
    for o in outs:
        if world in (4, 8):      # k = 10 (and for N = 4 also k = 8, 9): 64 N divides n; by column the windows are still exchanged, the pieces stay complete
            assert o["modes"]["proofs_row_sharded"] >= 1 and (o["modes"]["proofs_pieces_sharded"] >= 1) == (mode == "points"), o["modes"]
        assert o["shard_mode"] == mode
        for key, hexs in ref.items():
            assert o[key]["native"] == hexs and o[key]["python"] == hexs, key


def test_all_gather_path_and_emulated_all_to_all(zk, tmp_path):
    """the two fall-backs of the row-sharded exchange give the same bytes: complete columns all-gathered (option row_sharded = 0, what
    N = 3 / 5 and the extended-domain circuits use anyway), and the all-to-all emulated through the host transport's all-gather callback"""
    ref = _single_gpu_proofs(zk)
    vol = {}
    for extra in ({"ZKHIP_ROW_SHARDED": "0"}, {"ZKHIP_HOST_A2A": "0"}):
        outs = _run_workers(tmp_path, 2, True, 0, extra_env=extra)
        for o in outs:
            for key, hexs in ref.items():
                assert o[key]["native"] == hexs and o[key]["python"] == hexs, (extra, key)
        vol[next(iter(extra))] = outs[0]["small10evm"]["bytes_gathered"]
    # and the volumes differ as designed: the row-sharded exchange of the k = 10 circuit (the second run: row windows, the all-to-all emulated — the counter
    # reports what an all-to-all moves) receives less than the all-gathers of complete columns (the first run)
    assert vol["ZKHIP_HOST_A2A"] < vol["ZKHIP_ROW_SHARDED"], vol


def test_column_round_robin_sharding(zk, tmp_path):
    """the other MSM split (for k <= 19): every rank holds the WHOLE window tables and commits columns r, r + N, ... of a
    batch completely; the 96-byte results are all-gathered (non-owners contribute the identity).  Same bytes as the single-GPU proof."""
    ref = _single_gpu_proofs(zk)
    env_extra = {"ZK_SHARD_MODE": "columns"}
    os.environ.update(env_extra)
    try:
        outs = _run_workers(tmp_path, 2, True, 29641)
    finally:
        for k_ in env_extra:
            os.environ.pop(k_, None)
    for o in outs:
        assert o["shard_mode"] == "columns"
        for key, hexs in ref.items():
            assert o[key]["native"] == hexs and o[key]["python"] == hexs, key


# ---- the BASELINE configurations at FULL size over two ranks ("configs not exercised on the hardware they name" — the
# sharded path had only ever run at k <= 9).  One device, host-staged transport: c = 17 shard tables of 2^21 points, 384 MiB gathers,
# padded rounds at real size.  The single-GPU reference proof is made first and its memory given back before the ranks start.
def _single_then_sharded(zk, tmp_path, spec, make_shape, mode, timeout, extra_env=None, world=2, witness=1):
    import torch

    ffi, ctx = zk
    p = pv.Prover(pv.GpuBackend(ctx, ffi), make_shape(), satisfiable=True)
    w = p.witness(witness)
    ref = p.prove_native(w, transcript=spec[2])["proof"]
    p.release()
    p.b.params.free()
    del p, w
    ctx.trim()
    torch.cuda.empty_cache()
    outs = _run_workers(tmp_path, world, True, 0, shapes=[spec], extra_env={"ZK_SHARD_MODE": mode, "ZK_NATIVE_ONLY": "1", "ZK_WITNESS": str(witness), **(extra_env or {})}, timeout=timeout)
    key = f"{spec[0]}{spec[1]}{spec[2]}"
    for o in outs:
        assert o["shard_mode"] == mode and o["comm"]["nranks"] == world and o["comm"]["transport"] == (extra_env or {}).get("ZKHIP_COMM_TRANSPORT", "host")
        assert o[key]["native"] == ref.hex(), f"{key}: the sharded proof differs from the single-GPU proof"
    return ref, outs


def test_agg_k22_proof_over_two_ranks_by_point_range(zk, tmp_path):
    """BASELINE configs[3] at its real size: the aggregation-shaped k = 22 proof (Keccak) over 2 ranks by point range == the single-GPU
    proof bytes.  Also pins the exchange volume the sharding promises: no rank ever receives a complete extended column."""
    spec = ["agg", 22, "evm"]
    ref, outs = _single_then_sharded(zk, tmp_path, spec, lambda: pv.CircuitShape.agg(22, 3, 1), "points", 2400)
    assert len(ref) > 1000
    n = 1 << 22
    for o in outs:
        # What one rank receives per proof.  Round 2 all-gathered every complete coset column (5 advice / instance + 2 permuted + 5 products) and
        # the numerator: 13 x 384 MiB, half of it received at N = 2 = 2.6 GB.  Row-sharded: per phase an all-to-all of row windows (own range
        # + halo) of the columns the peer transformed — ceil(5/2) + ceil(2/2) + ceil(5/2) = 7 column-windows of 3 x n/2 rows — plus the
        # numerator's row ranges (3 x n/2 rows) and the latency-sized partial sums: 8 x 3 x n/2 x 32 B = 1.6 GB.
        # (round 3, later) the quotient's pieces stay row ranges: the numerator's all-gather (3 x n/2 rows) is replaced by two all-to-alls of
        # the blocks' row ranges to / from their owners (2 blocks of n/2 rows each way at N = 2, padded): 7 x 3 + 2 x 2 windows of n/2 rows
        # (and later still) the new columns live complete on their owner and as row ranges everywhere: + the coefficient forms' row ranges
        # from the owners (12 columns) and the z columns' Lagrange rows to them — at N = 2 that ADDS 0.33 GB (it pays from N = 4 on, where it
        # removes the replicated inverse transforms and grand products): (7 x 3 + 4 + 6) x n/2 x 32 B = 2.1 GB, against round 2's 2.6
        assert 0 < o["agg22evm"]["bytes_gathered"] <= (7 * 3 + 4 + 6) * (n // 2) * 32 + (32 << 20), o["agg22evm"]["bytes_gathered"]
        assert o["modes"] == {"proofs_row_sharded": 1, "proofs_pieces_sharded": 1, "shplonk_row_sharded": 1}, o["modes"]


def test_rsa_k17_proof_over_two_ranks_equals_the_cpu_oracle(zk, tmp_path, cpu_rsa17_proof):
    """BASELINE configs[1] (RSA k = 17, Poseidon) over 2 ranks by column AND by point range: both equal the single-GPU proof, which equals the CPU
    oracle backend's proof byte for byte (north_star: bit-identical to the CPU prover on the same SRS and witness; the oracle's proof of witness 0
    is made once per session: conftest.py)."""
    spec = ["rsa", 17, "poseidon"]
    ref, _ = _single_then_sharded(zk, tmp_path, spec, lambda: pv.CircuitShape.rsa(17), "columns", 900, witness=0)
    assert cpu_rsa17_proof == bytes(ref)
    (tmp_path / "points").mkdir()
    ref2, _ = _single_then_sharded(zk, tmp_path / "points", spec, lambda: pv.CircuitShape.rsa(17), "points", 900, witness=0)
    assert bytes(ref2) == bytes(ref)


def test_sha_k19_proof_over_two_ranks_by_column(zk, tmp_path):
    """BASELINE configs[2] at its real size (32 advice / 12 fixed columns, degree 5, Poseidon) over 2 ranks, MSMs split by column with whole
    window tables on both ranks == the single-GPU proof bytes"""
    spec = ["shafull", 19, "poseidon"]
    _single_then_sharded(zk, tmp_path, spec, lambda: pv.CircuitShape.sha256(19, n_advice=32, n_fixed=12), "columns", 1800)


# ---- the RCCL transport's own code path with more than one rank (a one-GPU box cannot run it against the real library: RCCL refuses two
# ranks per device).  tests/fake_rccl is a stand-in with the same entry points that moves the bytes through shared memory and CHECKS that
# every grouped send meets a receive of the same size on the peer (where the real library would hang).
def _fake_rccl():
    src = os.path.join(ROOT, "tests", "fake_rccl", "fake_rccl.cpp")
    lib = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")
    if not os.path.exists(lib):      # __graft_entry__.build() compiles it from the current source every time (file times do not survive the trip to the GPU box)
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-shared", "-fPIC", "-O1", src, "-o", lib])
    return lib


@pytest.mark.parametrize("world,mode", [(2, "points"), (4, "points"), (8, "points"), (2, "columns")])      # (4, "columns") took 29-44 s of the suite's budget: two ranks exercise the same branch
def test_rccl_transport_path_with_emulated_rccl(zk, tmp_path, world, mode):
    """zkhip_comm_init (dlopen, ncclCommInitRank, the all-to-all self-check), the event-fenced ncclAllGather and the grouped ncclSend / ncclRecv
    exchanges with their silent pairs — comm.hip's RCCL branch — under 2, 4 and 8 ranks: same proof bytes as the single GPU, the
    communicator reports transport "rccl" with ncclCommCount = N, the self-check passed (the row-sharded modes were taken), and no send
    was left without its receive (the stand-in turns that into an error)."""
    ref = _single_gpu_proofs(zk)
    # (four ranks: the library's own schedule only — the Python schedule's many small collectives through the stand-in's spin-waits made this case take 13 to 58 s
    # from run to run; it goes through the RCCL branch with 2 and 8 ranks)
    outs = _run_workers(tmp_path, world, True, 0, extra_env={"ZKHIP_RCCL_LIB": _fake_rccl(), "ZKHIP_COMM_TRANSPORT": "rccl", "ZK_SHARD_MODE": mode,
                                                            "ZKFAKE_RCCL_SLOT_MB": "8", **({"ZK_NATIVE_ONLY": "1"} if world == 4 else {})})
    for o in outs:
        assert o["transport"] == "rccl" and o["comm"]["transport"] == "rccl" and o["comm"]["transport_ranks"] == world and o["comm"]["nranks"] == world
        assert o["bytes_gathered"] > 0 and o["modes"]["proofs_row_sharded"] >= 1, o["modes"]
        if mode == "points":
            assert o["modes"]["proofs_pieces_sharded"] >= 1 and o["modes"]["shplonk_row_sharded"] >= 1, o["modes"]
        for key, hexs in ref.items():
            assert o[key]["native"] == hexs and o[key]["python"] == hexs, key


def test_a_stuck_collective_fails_the_host_wait_at_the_deadline(zk, tmp_path):
    """comm_timeout_ms (common.hpp wait_poll): the stand-in library leaves rank 0's collective stream spinning after the first all-gather that
    follows zkhip_comm_init (the worker arms ZKFAKE_RCCL_STALL=device:0:1 after the init) — a collective whose peer never arrives, as the host sees it.  Rank 0's next
    wait on its context fails after 2 s with the rank, the collective count and the phase in the message instead of polling for ever;
    rank 1, whose all-gather completed, is not affected."""
    outs = _run_workers(tmp_path, 2, True, 0, extra_env={"ZKHIP_RCCL_LIB": _fake_rccl(), "ZKHIP_COMM_TRANSPORT": "rccl", "ZKFAKE_RCCL_SLOT_MB": "8",
                                                         "ZK_STALL_TEST": "1", "ZKFAKE_RCCL_STALL_S": "40",
                                                         "ZKHIP_COMM_TIMEOUT_MS": "2000"}, timeout=300)
    assert outs[1]["error"] is None and outs[1]["recv"] == [1, 2]
    err = outs[0]["error"]
    assert err and "rank 0 of 2 stuck after collective #1" in err and "a host wait exceeded 2000 ms" in err, outs[0]
    # ... and the context can still be dropped: dead across zkhip_comm_destroy, later waits fail at once, close() returns (the stall lasts 40 s)
    assert outs[0]["ctx_dead"] == 1 and outs[0]["ctx_dead_after_comm_destroy"] == 1 and "taken for dead" in outs[0]["wait_after_destroy"], outs[0]
    assert outs[0]["close_s"] < 5.0, outs[0]


def test_a_stuck_collective_inside_a_proof_with_host_inputs_returns_at_the_deadline(zk, tmp_path):
    """ADVICE r4 (prover.hip StreamGuard): rank 0's collective stream stalls (40 s) after the last exchange of a proof whose advice columns are
    host arrays.  zkhip_create_proof_ex returns the deadline's error after ~2 s — its error exit does NOT wait for the streams the deadline
    has just declared stuck; rank 1, whose exchange was served, completes its proof."""
    outs = _run_workers(tmp_path, 2, True, 0, extra_env={"ZKHIP_RCCL_LIB": _fake_rccl(), "ZKHIP_COMM_TRANSPORT": "rccl", "ZKFAKE_RCCL_SLOT_MB": "8",
                                                         "ZK_STALL_TEST": "proof", "ZKFAKE_RCCL_STALL_S": "40",
                                                         "ZKHIP_COMM_TIMEOUT_MS": "2000"}, timeout=300)
    err = outs[0]["error"]
    assert err and "a host wait exceeded 2000 ms" in err and "rank 0 of 2 stuck" in err, outs[0]
    assert outs[0]["elapsed_s"] < 20.0 and outs[0]["close_s"] < 5.0, outs[0]
    assert "given up on by an earlier host wait" in outs[0]["second_call"] and outs[0]["second_call_s"] < 2.0, outs[0]
    assert outs[1]["error"] is None, outs[1]


def test_agg_k22_proof_over_two_ranks_through_the_rccl_branch(zk, tmp_path):
    """the full-size k = 22 proof over 2 ranks once more, through comm.hip's RCCL branch against the checking stand-in (600 MB all-to-all
    blocks through 2.6 GB of shared memory): the single-GPU proof's bytes, every send paired with its receive"""
    spec = ["agg", 22, "evm"]
    _, outs = _single_then_sharded(zk, tmp_path, spec, lambda: pv.CircuitShape.agg(22, 3, 1), "points", 2400,
                                   extra_env={"ZKHIP_RCCL_LIB": _fake_rccl(), "ZKHIP_COMM_TRANSPORT": "rccl", "ZKFAKE_RCCL_SLOT_MB": "700"})
    for o in outs:
        assert o["comm"]["transport_ranks"] == 2 and o["modes"] == {"proofs_row_sharded": 1, "proofs_pieces_sharded": 1, "shplonk_row_sharded": 1}


def test_agg_k20_proof_over_eight_ranks(zk, tmp_path):
    """the target node's rank count at a size where it matters: the aggregation-shaped k = 20 proof over EIGHT ranks by point range (c = 17-sized
    shard tables of 2^17 points, 4 MiB row windows; eight k = 22 ranks do not fit one device) == the single-GPU bytes, every exchange mode taken"""
    spec = ["agg", 20, "evm"]
    _, outs = _single_then_sharded(zk, tmp_path, spec, lambda: pv.CircuitShape.agg(20, 3, 1), "points", 1200, world=8)
    for o in outs:
        assert o["modes"] == {"proofs_row_sharded": 1, "proofs_pieces_sharded": 1, "shplonk_row_sharded": 1}
