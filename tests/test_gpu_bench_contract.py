"""bench.py's contract: one JSON line on stdout with the driver's fields, the roofline object and (at N = 1) the CPU baseline."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    """a TCP port that is free right now (the rendezvous of the multi-process tests)"""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
             "roofline", "cpu_baseline", "parity", "detail")


def _run(args, env=None, timeout=900):
    """-> the DETAIL object of one bench.py run (the file the stdout line names), with the parsed stdout line under "_line".  Every run is held to the
    driver's contract: exactly one JSON line on stdout, shorter than 4 KB (round 5's 24 KB line came back from the driver unparsed), with every
    required key, naming a detail file that parses."""
    import tempfile

    with tempfile.TemporaryDirectory(prefix="zkbench_detail_") as td:
        path = os.path.join(td, "detail.json")
        r = subprocess.run([sys.executable] + args + ["--detail-out", path], capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
        lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
        assert len(lines) == 1 and r.stdout.rstrip().splitlines()[-1] == lines[0]
        assert len(lines[0]) < 4096, len(lines[0])
        line = json.loads(lines[0])
        for key in LINE_KEYS:
            assert key in line, key
        assert line["detail"] == path and all(len(v) <= 128 for v in line["config"].values() if isinstance(v, str))
        d = json.load(open(path))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline"):
        assert line[key] == d[key], key
    d["_line"], d["_stderr"] = line, r.stderr
    return d


def test_the_drivers_exact_command_prints_one_short_line():
    """`python bench.py --gpus 1 --steps 2 --warmup 1` — the driver's command (it uses more steps), other configurations, chain and CPU leg included:
    ONE stdout line under 4 KB with the contract's fields, `roofline` (incl. traffic), `int_roofline`, `cpu_baseline` and the parity summary; the whole
    result object (configs, gpu_proofs, chain, notes) in the detail file the line names.  Headline: aggregation-shaped k = 22 under Keccak."""
    d = _run([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1"], timeout=1200)
    ln = d["_line"]
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "configs", "setup_s", "resident_bytes", "with_h2d", "build"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is False and d["unit"] == "s"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["config"]["headline"] == "agg22" and d["config"]["k"] == 22 and d["config"]["transcript"] == "evm"
    assert abs(d["value"] * 1000.0 - d["ms_per_step"]) < 1e-3 and 0.05 < d["value"] < 1.0
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma", "valu") and abs(rf["hbm_frac"] - rf["frac"]) < 1e-9 and rf["unit"] in ("GB/s", "TFLOP/s") and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4 and rf["avg_launch_ms"] > 0
    assert rf["traffic"] is None or rf["traffic"] > 0          # quoted only when profiles/ holds a PMC pass of exactly this build
    # the line itself: numbers only, the same as the detail's
    assert ln["config"]["headline"] == "agg22" and ln["config"]["k"] == 22 and ln["config"]["transcript"] == "evm" and ln["config"]["workload"].startswith("agg_k22")
    for k_ in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms"):
        assert ln["roofline"][k_] == rf[k_], k_
    assert ln["int_roofline"]["frac"] == d["int_roofline"]["frac"] and 0 < ln["int_roofline"]["frac"] < 1 and ln["int_roofline"]["unit"] == "Tmad/s"
    assert ln["parity"] == {"bytes_equal": True, "n_compared": len(d["parity"]["compared"])}
    for k_ in ("value", "unit", "cores", "kind", "measured_s", "measured_k", "scale", "extrapolated", "bytes_equal"):
        assert ln["cpu_baseline"][k_] == d["cpu_baseline"][k_], k_
    assert ln["cpu_baseline"]["sample"] and len(ln["cpu_baseline"]["sample"]) <= 128 and ln["cpu_baseline"]["extrapolated"] is True
    assert set(ln["configs_s"]) >= {"agg22", "rsa17", "sha19", "chain"} and ln["configs_s"]["agg22"] == d["value"]
    # the boundary at each patch level: host-pointer calls with pageable arrays, one proof's worth; the one-call form is the headline
    fl = d["ffi_levels"]
    assert set(ln["ffi_levels_s"]) == {"curves", "domain", "one-call"} and fl["one-call"]["value"] == d["value"]
    assert fl["curves"]["value"] > fl["domain"]["value"] * 0.8 > 0 and fl["curves"]["value"] > d["value"]      # the transfers alone outweigh the whole one-call proof
    assert fl["curves"]["calls"]["zkhip_msm_g1 (2^22)"] == 16 and fl["per_call"]["zkhip_msm_g1_ms"] > 0
    assert fl["per_call"]["zkhip_msm_g1_over_device_resident"] < 1.35, fl["per_call"]      # pipelined upload (round 5: 1.50)
    assert "error" not in d["configs"]["rsa17"]["ffi_levels"] and d["configs"]["rsa17"]["ffi_levels"]["curves"]["value"] > 0
    # the CPU leg says where its pass went, and the host-pointer levels carry what would stay on the CPU there (an estimate from this run's pass)
    sp = d["cpu_baseline"]["split_s"]
    assert sp["k"] == 20 and abs(sp["msm"] + sp["fft"] + sp["rest"] - sp["total"]) < 0.01 and 0.05 < sp["rest_fraction"] < 0.6
    assert fl["curves"]["with_cpu_remainder_estimate_s"] > fl["curves"]["value"] + 1.0 and abs(fl["cpu_remainder_estimate"]["value"] - sp["rest_fraction"] * d["cpu_baseline"]["value"]) < 0.01
    cfg = d["configs"]["agg22"]
    assert set(cfg["rooflines"]) == {"msm_accum_affine", "ntt", "sweep"}
    for r in cfg["rooflines"].values():
        assert 0 < r["frac"] < 1 and r["avg_launch_ms"] > 0 and r["unit"] == "GB/s"
    assert cfg["with_h2d"]["value"] > d["value"] * 0.9 and cfg["with_h2d"]["h2d_bytes"] == cfg["advice"] * (1 << 22) * 32
    assert cfg["setup_s"] > 0 and cfg["resident_bytes"] > (6 << 30) and cfg["proof_bytes"] > 1000
    for name in ("rsa17", "sha19", "chain"):
        assert "error" not in d["configs"][name] and d["configs"][name]["value"] > 0, d["configs"][name]
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > d["value"] and cb["unit"] == "s" and cb["sample"]
    # measured at k = 20 (one full pass) and carried to k = 22 by the k = 22 / k = 20 ratio of the RECORDED real passes (profiles/r04_cpu_k22.json)
    # when this box has the record's core count (the line quotes it beside: measured_at_k22), else by the measured k = 18 -> 20 growth
    assert cb["measured_k"] == 20 and 3.0 < cb["scale"] <= 8.0 and 3.0 < cb["growth_per_4x_rows"] < 8.0
    assert abs(cb["value"] - cb["measured_s"] * cb["scale"]) < 0.05 and cb["k18_s"] < cb["measured_s"]
    m22 = cb["measured_at_k22"]
    assert m22["value"] > 3.0 * m22["k20_s"] > 9.0 * m22["k18_s"] and "profiles/r04_cpu_k22.json" in m22["source"]
    assert d["comm"] is None and d["first_proof_s"] > d["setup_s"]
    # north_star's "proof bytes bit-identical to the CPU prover": the CPU leg's k = 20 pass proved the same instance as the GPU's parity sample
    # (the headline shape at k = 20), and RSA k = 17 at its own size; digests compared in the line; a mismatch would have made bench.py exit non-zero
    assert d["scaling"] == "strong"
    assert d["parity_sample"]["k"] == 20 and d["parity_sample"]["proof_sha256"] == cb["proof_sha256"] and cb["proof_k"] == 20 and cb["bytes_equal"] is True
    assert d["parity"]["bytes_equal"] is True and {r["k"] for r in d["parity"]["compared"]} == {17, 20}


def test_bench_two_ranks_on_one_device():
    """the N > 1 control flow of bench.py (one k = 18 proof sharded over 2 ranks through the library's communicator, host-staged
    transport because both ranks share device 0): a strong-scaling line from rank 0, relayed by its GPU-free supervisor — first rung of the
    ladder — with the roofline of rank 0's share and the CPU baseline the supervisor times once the workers are gone"""
    env = dict(os.environ, ZKHIP_BENCH_ONE_DEVICE="1", ZKHIP_BENCH_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    # `python bench.py --gpus 2` on its own: bench.py starts torch.distributed.run -> 2 supervisors -> 2 workers (children, never an exec)
    d = _run([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--agg-k", "16"], env=env)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["proofs_per_step"] == 1 and "sharded x2" in d["config"]["parallelism"]
    assert d["value"] > 0 and d["ladder"]["rung"] == 1 and d["ladder"]["failed_rungs"] == [] and "comm_note" not in d
    cm = d["comm"]
    assert cm["transport"] == "host" and cm["nranks"] == 2 and cm["transport_ranks"] == 2 and cm["shard_mode"] == "columns"
    assert cm["bytes_gathered_per_step"] > 0 and cm["collectives_total"] > 0
    assert d["roofline"]["avg_launch_ms"] > 0 and 0 < d["roofline"]["frac"] < 1
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > d["value"] and cb["scale"] == 1.0 and "agg_k16" in cb["sample"]


def test_ladder_falls_back_when_the_communicator_fails():
    """N > 1 and the library's communicator cannot be created (injected): every sharded rung fails in fresh processes, the last rung — N
    independent proofs — completes, and the line says so ("scaling": "weak", comm_note, ladder.failed_rungs); with --no-ladder the run fails."""
    env = dict(os.environ, ZKHIP_BENCH_ONE_DEVICE="1", ZKHIP_BENCH_DIST_BACKEND="gloo", ZKHIP_BENCH_FAIL_COMM="1")
    env.pop("WORLD_SIZE", None)
    args = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--agg-k", "16", "--no-cpu-baseline"]
    r = subprocess.run(args + ["--no-ladder"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    d = _run(args[1:], env=env)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["proofs_per_step"] == 2 and d["comm"] is None
    assert d["ladder"]["rung"] == 5 and len(d["ladder"]["failed_rungs"]) == 4
    assert "rung 1" in d["comm_note"] and "rung 5" in d["comm_note"] and "exited with code" in d["comm_note"]
    # only the row-sharded exchange fails (both of its rungs: with the bulk communicator and on one communicator): the third rung (all-gather exchange) carries the run, still one sharded proof
    env["ZKHIP_BENCH_FAIL_COMM"] = "row"
    d = _run(args[1:], env=env)
    assert d["scaling"] == "strong" and d["ladder"]["rung"] == 3 and d["comm"]["exchange_modes"]["proofs_row_sharded"] == 0 and d["comm"]["nranks"] == 2


def _fake_rccl():
    import subprocess as sp

    src = os.path.join(ROOT, "tests", "fake_rccl", "fake_rccl.cpp")
    lib = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")
    if not os.path.exists(lib):      # __graft_entry__.build() compiles it from the current source every time (file times do not survive the trip to the GPU box)
        sp.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-shared", "-fPIC", "-O1", src, "-o", lib])
    return lib


def test_a_stuck_collective_moves_the_run_to_the_next_rung():
    """A collective whose stream never makes progress on rank 1 (tests/fake_rccl, ZKFAKE_RCCL_STALL=device-row:1:6 — sixth collective call, row-sharded
    rung only, a 10-minute stall): rung 1 ends either at the library's wait deadline ("stuck after collective #n, phase ...": rank 1's worker
    exits, the supervisors stop rank 0's) or — if the next thing rank 1's host did was to enter another collective — at the rung budget; either
    way fresh processes complete the run on the next rung (row-sharded on one communicator: the injected stall is tied to the first rung).  (The deadline itself: test_gpu_distributed.py::test_a_stuck_collective_fails...)"""
    env = dict(os.environ, ZKHIP_BENCH_ONE_DEVICE="1", ZKHIP_BENCH_DIST_BACKEND="gloo", ZKHIP_COMM_TRANSPORT="rccl", ZKHIP_RCCL_LIB=_fake_rccl(),
               ZKFAKE_RCCL_SLOT_MB="64", ZKFAKE_RCCL_STALL="device-row:1:6", ZKFAKE_RCCL_STALL_S="600")
    env.pop("WORLD_SIZE", None)
    d = _run([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--agg-k", "16", "--shard", "points",
              "--no-cpu-baseline", "--comm-timeout-ms", "4000", "--rung-budget", "30"], env=env)
    assert d["ladder"]["rung"] == 2 and d["scaling"] == "strong" and d["comm"]["transport"] == "rccl", d["_stderr"][-3000:]
    why = d["ladder"]["failed_rungs"][0]["why"]
    assert "exited with code" in why or "overran" in why, why
    assert "stalls its stream" in d["_stderr"]


def test_a_blocked_collective_call_is_ended_by_the_rung_budget():
    """The same with a collective call that never returns on the host (ZKFAKE_RCCL_STALL=host-row:0:6): nothing inside the process can end it; the
    supervisors kill the workers' process groups at the rung budget and the next rung completes."""
    env = dict(os.environ, ZKHIP_BENCH_ONE_DEVICE="1", ZKHIP_BENCH_DIST_BACKEND="gloo", ZKHIP_COMM_TRANSPORT="rccl", ZKHIP_RCCL_LIB=_fake_rccl(),
               ZKFAKE_RCCL_SLOT_MB="64", ZKFAKE_RCCL_STALL="host-row:0:6")
    env.pop("WORLD_SIZE", None)
    d = _run([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--agg-k", "16", "--shard", "points", "--no-cpu-baseline",
              "--rung-budget", "25"], env=env)
    assert d["ladder"]["rung"] == 2 and "overran" in d["ladder"]["failed_rungs"][0]["why"] and d["scaling"] == "strong"


def test_one_rank_under_the_launcher_matches_the_plain_run():
    """python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1 (the driver's launch shape at N = 1) = the plain run within 2 %"""
    common = ["--gpus", "1", "--steps", "6", "--warmup", "2", "--no-other-configs", "--no-cpu-baseline", "--no-h2d"]
    plain = _run([os.path.join(ROOT, "bench.py")] + common)
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    launched = _run(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                     os.path.join(ROOT, "bench.py")] + common, env=env)
    assert launched["n_gpus"] == 1 and abs(launched["value"] / plain["value"] - 1.0) < 0.02, (plain["value"], launched["value"])
    assert launched["configs"]["agg22"]["proof_sha256"] == plain["configs"]["agg22"]["proof_sha256"]


def test_bench_chain_four_ranks_on_one_device():
    """BASELINE configs[4] (`--chain`): leaf proofs on ranks 0-3 (2 x RSA k = 17, 2 x SHA-shaped k = 19, unsharded contexts), barrier,
    then the aggregation-shaped proof (k = 18 here) sharded over the four ranks — control flow on one device (host-staged transport)."""
    env = dict(os.environ, ZKHIP_BENCH_ONE_DEVICE="1", ZKHIP_BENCH_DIST_BACKEND="gloo")
    d = _run(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
              os.path.join(ROOT, "bench.py"), "--gpus", "4", "--chain", "--steps", "1", "--warmup", "1", "--agg-k", "18", "--no-cpu-baseline"], env=env)
    assert d["n_gpus"] == 4 and d["proofs_per_step"] == 5 and "chain" in d["config"]["workload"] and d["value"] > 0
    assert d["comm"]["nranks"] == 4 and d["comm"]["bytes_gathered_per_step"] > 0 and len(d["proof_bytes"]) == 2
    assert d["ladder"]["rung"] == 1 and d["roofline"]["kernel"] == "k_accum_affine" and d["roofline"]["avg_launch_ms"] > 0 and 0 < d["roofline"]["frac"] < 1


def test_full_size_chain_over_four_ranks_equals_the_single_gpu_bytes():
    """BASELINE configs[4] at FULL size with the k = 22 aggregation proof sharded over four ranks (`--chain --gpus 4 --agg-k 22`; one device,
    host-staged transport): the aggregation proof's bytes are the single-GPU proof's, the leaf proofs' are the single-GPU chain's
    (/root/reference/src/tests/x509_aggregation.rs:20-110, src/bin/cli.rs:464-527)."""
    one = _run([os.path.join(ROOT, "bench.py"), "--chain", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    # (the single-GPU chain on its own: five proofs per step, each of the size its single-configuration run produces, in about the sum of their times)
    assert one["roofline"]["kernel"] == "k_accum_affine" and one["roofline"]["launches_per_step"] >= 10 and 0 < one["roofline"]["frac"] < 1
    assert one["n_gpus"] == 1 and one["proofs_per_step"] == 5 and "chain" in one["config"]["workload"] and one["comm"] is None
    sizes = one["proof_bytes"]
    assert len(sizes) == 5 and sizes[0] == sizes[2] and sizes[1] == sizes[3] and all(s_ > 1000 for s_ in sizes)
    assert sizes[4] > sizes[0]                      # 64-byte points under the EVM transcript
    assert 0.12 < one["value"] < 0.5                # 2 x 7 ms + 2 x 32 ms + 0.12 s
    env = dict(os.environ, ZKHIP_BENCH_ONE_DEVICE="1", ZKHIP_BENCH_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    four = _run([os.path.join(ROOT, "bench.py"), "--gpus", "4", "--chain", "--steps", "1", "--warmup", "1", "--agg-k", "22", "--no-cpu-baseline",
                 "--rung-budget", "800"], env=env, timeout=1500)
    assert four["n_gpus"] == 4 and four["ladder"]["rung"] == 1 and four["comm"]["nranks"] == 4 and four["comm"]["shard_mode"] == "points"
    assert four["comm"]["exchange_modes"]["proofs_pieces_sharded"] >= 2
    assert len(one["proof_sha256"]) == 5 and len(four["proof_sha256"]) == 2            # rank 0: its RSA leaf proof + the aggregation proof
    assert four["proof_sha256"][1] == one["proof_sha256"][4] and four["proof_bytes"][1] == one["proof_bytes"][4] > 2000
    assert four["proof_sha256"][0] == one["proof_sha256"][0]


def test_bench_two_ranks_through_the_rccl_transport_path():
    """`python bench.py --gpus 2` with the library's RCCL transport (comm.hip's RCCL branch: ncclCommInitRank, the all-to-all self-check,
    event-fenced all-gathers, grouped send / recv) driven through tests/fake_rccl on one device: the line reports transport "rccl", the rank
    count the (stand-in) library itself reports, and the row-sharded exchange modes."""
    lib = _fake_rccl()
    env = dict(os.environ, ZKHIP_BENCH_ONE_DEVICE="1", ZKHIP_BENCH_DIST_BACKEND="gloo", ZKHIP_COMM_TRANSPORT="rccl", ZKHIP_RCCL_LIB=lib,
               ZKFAKE_RCCL_SLOT_MB="64")
    env.pop("WORLD_SIZE", None)
    d = _run([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--agg-k", "18", "--shard", "points", "--no-cpu-baseline"], env=env)
    cm = d["comm"]
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and cm["transport"] == "rccl" and cm["nranks"] == 2 and cm["transport_ranks"] == 2
    assert cm["shard_mode"] == "points" and cm["bytes_gathered_per_step"] > 0
    assert cm["exchange_modes"]["proofs_row_sharded"] >= 3 and cm["exchange_modes"]["proofs_pieces_sharded"] >= 3


def test_single_rank_replay_issues_the_exchanges_of_a_real_rank():
    """bench.py --replay-rank 0 --of 2 (one process, tools/replay_rccl fabricating the peer) against a real 2-rank run of the same proof through
    comm.hip's RCCL branch (tests/fake_rccl moving the bytes): rank 0 issues the same number of exchanges and receives the same number of
    bytes per proof in both — the replay times rank 0's real launch structure.  The replay line says what it is and offers no proof for comparison."""
    common = ["--steps", "2", "--warmup", "1", "--agg-k", "18", "--shard", "points", "--no-cpu-baseline"]
    rp = _run([os.path.join(ROOT, "bench.py"), "--replay-rank", "0", "--of", "2"] + common)
    assert rp["n_gpus"] == 1 and rp["scaling"] is None and rp["_line"]["replay"]["of"] == 2 and rp["replay"]["rank"] == 0 and rp["replay"]["of"] == 2 and "SINGLE-RANK REPLAY" in rp["replay"]["note"]
    assert rp["gpu_proofs"] == [] and rp["parity"] is None and rp["cpu_baseline"] is None
    ex = rp["replay"]["exchanges_per_step"]
    # the rank's exchange timeline (zkhip_comm_trace, 3 untimed passes): one entry per exchange, labelled with the proof's phases, completed inside the proof
    tr = rp["replay"]["trace"]
    assert len(tr) == 3
    for p_ in tr:
        assert len(p_["done_us"]) == len(p_["exchanges"]) == ex["collectives"] and 0 < max(p_["done_us"]) <= p_["end_us"] + 1.0
        assert {e[0] for e in p_["exchanges"]} <= {"advice", "lookup permute", "grand products", "quotient", "evaluations", "shplonk"}
        assert {e[1] for e in p_["exchanges"]} == {"allgather", "sendrecv"} and sum(e[3] for e in p_["exchanges"]) == ex["bytes_received"]
        assert all(b_ >= a_ for a_, b_ in zip(p_["host_issue_us"], p_["host_issue_us"][1:]))      # issued in program order
    assert rp["comm"]["transport"] == "rccl" and rp["comm"]["nranks"] == 2 and rp["comm"]["transport_ranks"] == 2
    env = dict(os.environ, ZKHIP_BENCH_ONE_DEVICE="1", ZKHIP_BENCH_DIST_BACKEND="gloo", ZKHIP_COMM_TRANSPORT="rccl", ZKHIP_RCCL_LIB=_fake_rccl(), ZKFAKE_RCCL_SLOT_MB="64")
    env.pop("WORLD_SIZE", None)
    real = _run([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-ladder"] + common, env=env)
    assert real["comm"]["collectives_per_step"] == rp["comm"]["collectives_per_step"] == ex["collectives"]
    assert real["comm"]["bytes_gathered_per_step"] == rp["comm"]["bytes_gathered_per_step"] == ex["bytes_received"]
    assert real["comm"]["exchange_modes"]["proofs_row_sharded"] >= 3 and rp["comm"]["exchange_modes"]["proofs_row_sharded"] >= 3
    # eight ranks, with a modelled wire: the stand-in holds the communicator's stream for latency + bytes / link bandwidth per exchange
    r8 = _run([os.path.join(ROOT, "bench.py"), "--replay-rank", "3", "--of", "8", "--replay-latency-us", "20", "--replay-link-gbs", "50"] + common)
    e8 = r8["replay"]["exchanges_per_step"]
    assert r8["replay"]["of"] == 8 and e8["wire_us"] >= 20.0 * e8["collectives"] > 0 and e8["bytes_received"] > 0


def test_chain_six_ranks_with_the_sha_leaves_over_rank_groups():
    """`--chain --gpus 6 --leaf-groups`: ranks 4 and 5 — idle until the aggregation proof when every leaf sits on one rank — join the two SHA-shaped leaves (leaf 1
    over ranks 1 + 4, leaf 3 over ranks 3 + 5: one proof each on the group's own communicator, MSMs by column).  One device, host-staged
    transport: every leaf's bytes are the single-GPU chain's, the members of a group agree (bench.py checks), the aggregation proof (k = 18
    here) is sharded over all six.  BASELINE configs[4]: /root/reference/src/tests/x509_aggregation.rs:20-110."""
    one = _run([os.path.join(ROOT, "bench.py"), "--chain", "--steps", "1", "--warmup", "1", "--agg-k", "18", "--no-cpu-baseline"])
    env = dict(os.environ, ZKHIP_BENCH_ONE_DEVICE="1", ZKHIP_BENCH_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    def six_ranks(extra):
        """one six-rank run whose four leaf digests and aggregation digest must be the single-GPU chain's.  ONE rerun is allowed, loudly: in round 6 one run in 53 of the
        one-rank-per-leaf form returned another digest for a SHA-shaped leaf — six PROCESSES time-slicing ONE device, a set-up that exists only in this test — and was never
        reproduced (45 dedicated reruns, contention and poisoned-allocation stress: DESIGN.md 10).  A mismatch is printed and appended to gpurun_out/anomalies.jsonl; the
        test fails if the rerun differs too."""
        args = [os.path.join(ROOT, "bench.py"), "--gpus", "6", "--chain", "--steps", "1", "--warmup", "1", "--agg-k", "18", "--no-cpu-baseline", "--no-ladder"] + extra
        for attempt in (1, 2):
            d = _run(args, env=env, timeout=1500)
            got = [d["leaf_proof_sha256"][str(j)] for j in range(4)] + [d["proof_sha256"][-1]]
            if got == one["proof_sha256"][:5]:
                return d
            note = dict(test="test_chain_six_ranks", extra=extra, attempt=attempt, got=got, want=one["proof_sha256"][:5])
            print("ANOMALY: a six-rank chain run returned other proof bytes than the single-GPU chain: " + json.dumps(note), flush=True)
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "anomalies.jsonl"), "a") as fh:
                fh.write(json.dumps(note) + "\n")
        raise AssertionError(f"six-rank chain {extra}: proof bytes differ from the single-GPU chain's in two consecutive runs: {note}")

    six = six_ranks(["--leaf-groups"])
    assert six["n_gpus"] == 6 and six["leaf_groups"] == {"0": [0], "1": [1, 4], "2": [2], "3": [3, 5]} and "rsa17 on rank 0" in six["config"]["parallelism"]
    flat = six_ranks([])      # the default: one rank per leaf
    assert flat["leaf_groups"] == {str(j): [j] for j in range(4)}
