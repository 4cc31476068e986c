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


ROW_WORKER = r'''
"""The arithmetic of the row-range exchanges of csrc/{shplonk,polyops,prover}.hip, restated with big integers over gloo: every rank holds rows
[R m, (R + 1) m) of the inputs and must end with ITS rows of the single-process result."""
import json, os, sys
sys.path[:0] = [os.environ["ZK_ROOT"], os.path.join(os.environ["ZK_ROOT"], "oracle"), os.path.join(os.environ["ZK_ROOT"], "tests")]
import torch.distributed as dist
import zkoracle_py as zo
R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n = 1 << 9
m = n // world
lo = rank * m
full = lambda seed: [zo.limbs_to_int(r_) % R for r_ in zo.synth_raw253(seed, n)]

def allgather(obj):
    out = [None] * world
    dist.all_gather_object(out, obj)
    return out

# ---- Kate division by (X - r): q[j] = s[j+1], s[j] = a[j] + r s[j+1].  Local pass with zero carry-in, range totals T_R = sum_j a_j r^j,
# carry-in c_R = T_(R+1) + r^m (T_(R+2) + ...), fix q[j] += c_R r^(m-1-j)     (shplonk.hip divide_round(.., sharded), k_kd_fix)
a, r = full(11), full(12)[5]
mine = a[lo:lo + m]
q0, s = [0] * m, 0
for j in range(m - 1, -1, -1):
    q0[j] = s
    s = (mine[j] + r * s) % R
tot = allgather(s)                              # s = the range's Horner total
c, rm = 0, pow(r, m, R)
for q_ in range(world - 1, rank, -1):
    c = (c * rm + tot[q_]) % R
kate = [(q0[j] + c * pow(r, m - 1 - j, R)) % R for j in range(m)]

# ---- grand products of two chained segments: z_s[i] = first_s * prod_{j < i} t_s[j], first_0 = 1, first_1 = z_0[last] (unchained prefix at
# row `last`); per rank: local exclusive prefixes, totals, the pick at `last` from the rank that holds it   (polyops.hip k_rp_shard_first)
last = n - 6
terms = [full(21), full(22)]
loc, tots, picks = [], [], []
for t in terms:
    pre, acc = [], 1
    for j in range(lo, lo + m):
        pre.append(acc)
        acc = acc * t[j] % R
    loc.append(pre)
    tots.append(acc)
    picks.append(pre[last - lo] if lo <= last < lo + m else 1)
got = allgather((tots, picks))
chain_rank = last // m
zs, f = [], 1
for s_ in range(2):
    below = 1
    for q_ in range(rank):
        below = below * got[q_][0][s_] % R
    zs.append([below * f % R * v % R for v in loc[s_]])
    full_last = got[chain_rank][1][s_]
    for q_ in range(chain_rank):
        full_last = full_last * got[q_][0][s_] % R
    f = f * full_last % R

# ---- h(x) from per-rank partial evaluations: h(x) = sum_R x^(R m) P_R(x)      (prover.hip, pieces on row ranges)
h, x = full(31), full(32)[7]
part = 0
for j in range(m - 1, -1, -1):
    part = (part * x + h[lo + j]) % R
parts = allgather(part)
hx, xm = 0, pow(x, m, R)
for q_ in range(world - 1, -1, -1):
    hx = (hx * xm + parts[q_]) % R
json.dump(dict(kate=[hex(v) for v in kate], z=[[hex(v) for v in z_] for z_ in zs], hx=hex(hx)), open(os.path.join(os.environ["ZK_OUT"], f"rows{rank}.json"), "w"))
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world", [2, 4])
def test_row_range_exchanges_match_the_oracle(tmp_path, oracle, world):
    """The three cross-rank recurrences the row-sharded proof relies on — Kate division with carries, grand products chained across ranks and
    sets, h(x) from partial evaluations — restated with integers over gloo: the ranks' rows, concatenated, equal the ORACLE's single-process
    results (zo.kate_division, running products, zo.eval_polynomial).  The C++ forms are checked byte for byte on the GPU
    (tests/test_gpu_distributed.py); this pins the arithmetic they implement on the CPU suite."""
    zo = oracle
    R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
    worker = tmp_path / "rows_worker.py"
    worker.write_text(ROW_WORKER)
    env = dict(os.environ, ZK_ROOT=ROOT, ZK_OUT=str(tmp_path), OMP_NUM_THREADS="1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
                           "127.0.0.1", "--master-port", str(_free_port()), str(worker)], env=env, timeout=600)
    outs = [json.loads((tmp_path / f"rows{r}.json").read_text()) for r in range(world)]
    n = 1 << 9
    full = lambda seed: [zo.limbs_to_int(r_) % R for r_ in zo.synth_raw253(seed, n)]
    arr = lambda ints: zo.fr_arr_from_ints(ints)
    a, r = full(11), full(12)[5]
    exp = zo.fr_arr_to_ints(zo.kate_division(arr(a), arr([r])))
    got = [int(v, 16) for o in outs for v in o["kate"]]
    assert got[:n - 1] == exp[:n - 1]                     # (the oracle keeps n - 1 quotient coefficients)
    terms = [full(21), full(22)]
    last, f = n - 6, 1
    for s_ in range(2):
        z, acc = [], f
        for j in range(n):
            z.append(acc)
            acc = acc * terms[s_][j] % R
        assert [int(v, 16) for o in outs for v in o["z"][s_]] == z
        f = z[last]
    h, x = full(31), full(32)[7]
    assert all(int(o["hx"], 16) == zo.fr_to_int(zo.eval_polynomial(arr(h), zo.fr_from_int(x))) for o in outs)
