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


def _run(args, env=None, timeout=900):
    r = subprocess.run([sys.executable] + args, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_bench_prints_one_contract_line():
    """the default headline (aggregation-shaped k = 22 under Keccak), one step, no other configurations"""
    d = _run([os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-other-configs"])
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
    cfg = d["configs"]["agg22"]
    assert set(cfg["rooflines"]) == {"msm_accum_affine", "ntt", "sweep"}
    for r in cfg["rooflines"].values():
        assert 0 < r["frac"] < 1 and r["avg_launch_ms"] > 0 and r["unit"] == "GB/s"
    assert cfg["with_h2d"]["value"] > d["value"] * 0.9 and cfg["with_h2d"]["h2d_bytes"] == cfg["advice"] * (1 << 22) * 32
    assert cfg["setup_s"] > 0 and cfg["resident_bytes"] > (6 << 30) and cfg["proof_bytes"] > 1000
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > d["value"] and cb["unit"] == "s" and cb["sample"]
    # measured at k = 20 (one full pass) and carried to k = 22 by the k = 22 / k = 20 ratio of the RECORDED real passes (profiles/r04_cpu_k22.json),
    # which the line quotes beside it (measured_at_k22: scale 1 on record)
    assert cb["measured_k"] == 20 and 3.0 < cb["scale"] <= 5.0 and 3.0 < cb["growth_per_4x_rows"] < 8.0
    assert abs(cb["value"] - cb["measured_s"] * cb["scale"]) < 0.05 and cb["k18_s"] < cb["measured_s"]
    m22 = cb["measured_at_k22"]
    assert m22["value"] > 3.0 * m22["k20_s"] > 9.0 * m22["k18_s"] and m22["cores"] == cb["cores"] and "profiles/r04_cpu_k22.json" in m22["source"]
    assert abs(cb["scale"] - m22["value"] / m22["k20_s"]) < 1e-3
    assert d["comm"] is None and d["first_proof_s"] > d["setup_s"]


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
    assert d["ladder"]["rung"] == 4 and len(d["ladder"]["failed_rungs"]) == 3
    assert "rung 1" in d["comm_note"] and "rung 4" in d["comm_note"] and "exited with code" in d["comm_note"]
    # only the row-sharded exchange fails: the second rung (all-gather exchange) carries the run, still one sharded proof
    env["ZKHIP_BENCH_FAIL_COMM"] = "row"
    d = _run(args[1:], env=env)
    assert d["scaling"] == "strong" and d["ladder"]["rung"] == 2 and d["comm"]["exchange_modes"]["proofs_row_sharded"] == 0 and d["comm"]["nranks"] == 2


def _fake_rccl():
    lib = _fake_rccl()
    env = dict(os.environ, ZKHIP_BENCH_ONE_DEVICE="1", ZKHIP_BENCH_DIST_BACKEND="gloo", ZKHIP_COMM_TRANSPORT="rccl", ZKHIP_RCCL_LIB=lib,
               ZKFAKE_RCCL_SLOT_MB="64")
    env.pop("WORLD_SIZE", None)
    d = _run([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--agg-k", "18", "--shard", "points", "--no-cpu-baseline"], env=env)
    cm = d["comm"]
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and cm["transport"] == "rccl" and cm["nranks"] == 2 and cm["transport_ranks"] == 2
    assert cm["shard_mode"] == "points" and cm["bytes_gathered_per_step"] > 0
    assert cm["exchange_modes"]["proofs_row_sharded"] >= 3 and cm["exchange_modes"]["proofs_pieces_sharded"] >= 3
