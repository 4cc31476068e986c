"""bench.py --gpus N: the GPU-free supervisors and their fallback ladder, exercised on the CPU with a stand-in worker
(tests/fake_bench_worker.py): a failing or hanging rank on one rung moves EVERY rank to the next rung in fresh processes; the line says
which rung it came from and why the earlier ones failed; the run fails only if every rung does.  The reference's multi-proof flow this
protects: /root/reference/src/tests/x509_aggregation.rs:20-110, src/bin/cli.rs:464-527."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _details(r, detail):
    """the detail object of every JSON line on stdout (each line is held to the driver's bound: under 4 KB, naming a detail file that parses)"""
    out = []
    for ln in r.stdout.splitlines():
        if ln.strip().startswith("{"):
            assert len(ln) < 4096, len(ln)
            line = json.loads(ln)
            assert line["detail"] == detail and line["ladder"]["rung"] >= 1 and "cpu_baseline" in line and "roofline" in line and "parity" in line
            d = json.load(open(detail))
            assert d["ladder"]["rung"] == line["ladder"]["rung"] and d["value"] == line["value"] and d["n_gpus"] == line["n_gpus"]
            out.append(d)
    return out


def _run(tmp_path, plan, extra=(), world=2, timeout=300):
    env = dict(os.environ, ZKHIP_BENCH_WORKER_SCRIPT=os.path.join(ROOT, "tests", "fake_bench_worker.py"), FAKE_PLAN=json.dumps(plan),
               FAKE_COUNT_DIR=str(tmp_path))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "ZKHIP_BENCH_ROLE"):
        env.pop(k, None)
    t0 = time.time()
    detail = os.path.join(str(tmp_path), "detail.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--no-cpu-baseline", "--detail-out", detail] + list(extra), capture_output=True,
                       text=True, timeout=timeout, cwd=ROOT, env=env)
    return r, _details(r, detail), time.time() - t0


def test_first_rung_succeeds(tmp_path):
    r, lines, _ = _run(tmp_path, {})
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1 and lines[0]["ladder"]["rung"] == 1 and lines[0]["ladder"]["failed_rungs"] == [] and "comm_note" not in lines[0]
    assert lines[0]["row_sharded_env"] == "1" and "--replicas" not in lines[0]["argv"]


def test_a_failing_rank_moves_every_rank_to_the_next_rung(tmp_path):
    r, lines, _ = _run(tmp_path, {"1": {"1": "fail"}})
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1
    d = lines[0]
    assert d["ladder"]["rung"] == 2 and d["row_sharded_env"] == "1" and d["comm_bulk_env"] == "0" and "ONE communicator" in d["rung_label"]      # still row-sharded: only the bulk communicator and torch's RCCL control plane are dropped
    assert len(d["ladder"]["failed_rungs"]) == 1 and "rank 1" in d["ladder"]["failed_rungs"][0]["why"] and "code 3" in d["ladder"]["failed_rungs"][0]["why"]
    assert "rung 1" in d["comm_note"] and "rung 2" in d["comm_note"]
    # every rank started exactly two workers: fresh processes per rung
    assert [open(os.path.join(tmp_path, f"rank{q}")).read() for q in (0, 1)] == ["2", "2"]


def test_a_hanging_rank_is_killed_at_the_rung_budget(tmp_path):
    r, lines, dt = _run(tmp_path, {"1": {"0": "hang"}, "2": {"1": "hang"}}, extra=["--rung-budget", "6"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = lines[0]
    assert d["ladder"]["rung"] == 3 and d["row_sharded_env"] == "0" and "all-gather" in d["rung_label"]
    assert "overran" in d["ladder"]["failed_rungs"][0]["why"] and "overran" in d["ladder"]["failed_rungs"][1]["why"]
    assert dt < 120


def test_last_rung_is_independent_proofs(tmp_path):
    r, lines, _ = _run(tmp_path, {str(i): {"0": "fail"} for i in (1, 2, 3, 4)})
    assert r.returncode == 0, r.stderr[-2000:]
    assert lines[0]["ladder"]["rung"] == 5 and "--replicas" in lines[0]["argv"] and len(lines[0]["ladder"]["failed_rungs"]) == 4
    assert "--shard" in lines[0]["ladder"]["failed_rungs"][3]["label"] or "by column" in lines[0]["ladder"]["failed_rungs"][3]["label"]


def test_every_rung_failing_fails_the_run(tmp_path):
    r, lines, _ = _run(tmp_path, {str(i): {"1": "fail"} for i in (1, 2, 3, 4, 5)})
    assert r.returncode != 0 and lines == []
    assert "every rung of the ladder failed" in r.stderr


def test_no_ladder_means_first_rung_only(tmp_path):
    r, lines, _ = _run(tmp_path, {"1": {"0": "fail"}}, extra=["--no-ladder"])
    assert r.returncode != 0 and lines == []


def test_chain_ladder_ends_with_the_unsharded_aggregation(tmp_path):
    r, lines, _ = _run(tmp_path, {str(i): {"3": "fail"} for i in (1, 2, 3, 4)}, extra=["--chain"], world=4)
    assert r.returncode == 0, r.stderr[-2000:]
    assert lines[0]["ladder"]["rung"] == 5 and "--agg-unsharded" in lines[0]["argv"]


def test_supervisors_under_the_drivers_own_launch_line(tmp_path):
    """python -m torch.distributed.run ... bench.py --gpus 2 (how the driver starts N > 1): the same ladder, no parent of ours"""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, ZKHIP_BENCH_WORKER_SCRIPT=os.path.join(ROOT, "tests", "fake_bench_worker.py"), FAKE_PLAN=json.dumps({"1": {"0": "fail"}}),
               FAKE_COUNT_DIR=str(tmp_path))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "ZKHIP_BENCH_ROLE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--detail-out", os.path.join(str(tmp_path), "detail.json")],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _details(r, os.path.join(str(tmp_path), "detail.json"))
    assert len(lines) == 1 and lines[0]["ladder"]["rung"] == 2


def test_the_librarys_transport_does_not_follow_torchs_backend(monkeypatch):
    """The ladder's later rungs put TORCH on gloo (ZKHIP_BENCH_DIST_BACKEND) so that the library's communicator is the only RCCL communicator of the process: the library's
    own exchanges stay on RCCL there.  Host staging only where RCCL cannot run (every rank on one device: the one-GPU tests) or where the environment asks for it."""
    sys.path.insert(0, ROOT)
    import bench

    for k in ("ZKHIP_COMM_TRANSPORT", "ZKHIP_BENCH_ONE_DEVICE", "ZKHIP_BENCH_DIST_BACKEND"):
        monkeypatch.delenv(k, raising=False)
    assert bench.lib_transport() == "rccl"
    monkeypatch.setenv("ZKHIP_BENCH_DIST_BACKEND", "gloo")
    assert bench.lib_transport() == "rccl"
    monkeypatch.setenv("ZKHIP_BENCH_ONE_DEVICE", "1")
    assert bench.lib_transport() == "host"
    monkeypatch.setenv("ZKHIP_COMM_TRANSPORT", "rccl")
    assert bench.lib_transport() == "rccl"
    for label, extra_args, extra_env in bench.LADDER["shard"][1:] + bench.LADDER["chain"][1:]:
        assert extra_env.get("ZKHIP_BENCH_DIST_BACKEND") == "gloo" and "ZKHIP_COMM_TRANSPORT" not in extra_env, label
