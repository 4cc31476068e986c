"""TEST INFRASTRUCTURE: stands in for bench.py's GPU worker rank in the CPU-only tests of the fallback ladder (tests/test_bench_ladder_cpu.py).
bench.py's supervisors start it instead of the real worker when ZKHIP_BENCH_WORKER_SCRIPT names it.  FAKE_PLAN is a JSON object
{"<rung index, 1-based>": {"<rank>": "fail" | "hang" | "ok"}}; anything not named is "ok".  An "ok" rank 0 prints a minimal JSON line."""
import json
import os
import sys
import time

rank = int(os.environ["RANK"])
assert os.environ.get("ZKHIP_BENCH_ROLE") == "worker" and "MASTER_PORT" in os.environ
label = os.environ["ZKHIP_BENCH_RUNG"]
plan = json.loads(os.environ.get("FAKE_PLAN", "{}"))
count_file = os.environ["FAKE_COUNT_DIR"] + f"/rank{rank}"
n = int(open(count_file).read()) + 1 if os.path.exists(count_file) else 1
open(count_file, "w").write(str(n))          # which rung this is for this rank: the number of workers started so far
what = plan.get(str(n), {}).get(str(rank), "ok")
if what == "fail":
    print(f"fake worker rank {rank}: injected failure on rung {n}", file=sys.stderr)
    sys.exit(3)
if what == "hang":
    time.sleep(3600)
time.sleep(0.5)
if rank == 0:
    print(json.dumps({"metric": "fake", "value": 1.0, "n_gpus": int(os.environ["WORLD_SIZE"]), "rung_label": label, "argv": sys.argv[1:], "cpu_baseline": None,
                      "row_sharded_env": os.environ.get("ZKHIP_ROW_SHARDED", "1"), "comm_bulk_env": os.environ.get("ZKHIP_COMM_BULK", "1"), "config": {"k": int(os.environ.get("FAKE_K", "16")), "transcript": "evm"},
                      "gpu_proofs": json.loads(os.environ.get("FAKE_GPU_PROOFS", "[]"))}))
