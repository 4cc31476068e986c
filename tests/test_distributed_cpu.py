"""CPU, world_size 2 over gloo: the point-range-sharded commitment path (ShardedCommit) gives the same
transcript as the single-process schedule.  Compute is the oracle backend (test infrastructure)."""
import json
import os
import subprocess
import sys


def _free_port():
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
sys.path[:0] = [os.environ["ZK_ROOT"], os.path.join(os.environ["ZK_ROOT"], "oracle"), os.path.join(os.environ["ZK_ROOT"], "tests")]
import torch.distributed as dist
import halo2_zkcert_amd.prover as pv
from oracle_backend import OracleBackend
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
sh = pv.CircuitShape.small(5)
p = pv.Prover(pv.ShardedCommit(OracleBackend(1), rank, world, dist, shard_ntt=os.environ.get("ZK_SHARD_NTT") == "1",
                              shard_sweep=os.environ.get("ZK_SHARD_SWEEP") == "1"), sh)
t = p.prove(p.witness(2))
out = dict(rank=rank, commitments=t["commitments"], challenges={k: hex(v) for k, v in t["challenges"].items()})
open(os.path.join(os.environ["ZK_OUT"], f"rank{rank}.json"), "w").write(json.dumps(out))
dist.barrier()
dist.destroy_process_group()
'''


import pytest


@pytest.mark.parametrize("shard_ntt,shard_sweep", [("0", "0"), ("1", "0"), ("1", "1")])
def test_sharded_commit_world2(tmp_path, oracle, shard_ntt, shard_sweep):
    """shard_ntt = 1 additionally distributes the coset NTTs by polynomial and all-gathers the extended columns; shard_sweep = 1
    evaluates the quotient sweep by row range (each rank half of the extended rows) and all-gathers h"""
    worker = tmp_path / "worker.py"
    worker.write_text(WORKER)
    env = dict(os.environ, ZK_ROOT=ROOT, ZK_OUT=str(tmp_path), OMP_NUM_THREADS="1", ZK_SHARD_NTT=shard_ntt, ZK_SHARD_SWEEP=shard_sweep)
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                           "127.0.0.1", "--master-port", str(_free_port()), str(worker)], env=env, timeout=600)
    r0 = json.loads((tmp_path / "rank0.json").read_text())
    r1 = json.loads((tmp_path / "rank1.json").read_text())
    assert r0["commitments"] == r1["commitments"] and r0["challenges"] == r1["challenges"]
    sys.path[:0] = [os.path.join(ROOT, "tests")]
    import halo2_zkcert_amd.prover as pv
    from oracle_backend import OracleBackend

    sh = pv.CircuitShape.small(5)
    p = pv.Prover(OracleBackend(2), sh)
    t = p.prove(p.witness(2))
    assert [list(c) for c in t["commitments"]] == r0["commitments"]
    assert {k: hex(v) for k, v in t["challenges"].items()} == r0["challenges"]
