"""bench.py's byte comparison of the HIP path's proofs with the CPU oracle's (north_star: "proof bytes bit-identical to the reference CPU prover
on the same SRS and witness"; the reference's proving calls: /root/reference/src/helpers.rs:233,299, src/bin/cli.rs:320,369,519).  CPU only:
the digest bookkeeping, and the supervisor's exit code when a digest differs (stand-in worker, tests/fake_bench_worker.py)."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [p for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")) if p not in sys.path]


def test_parity_check_rows():
    import bench

    cpu = [dict(shape="a", k=5, transcript="evm", witness=0, proof_sha256="11", proof_bytes=10),
           dict(shape="b", k=6, transcript="poseidon", witness=1, proof_sha256="22", proof_bytes=12)]
    g = lambda shape, kind, w, d: dict(shape=shape, k=5, transcript=kind, witness=w, proof_sha256=d, where="t")
    ok = bench.parity_check([g("a", "evm", 0, "11"), g("b", "poseidon", 1, "22"), g("c", "evm", 0, "99")], cpu)
    assert ok["bytes_equal"] is True and len(ok["compared"]) == 2                       # "c" has no CPU counterpart: not a row
    assert bench.parity_check([g("a", "evm", 0, "12")], cpu)["bytes_equal"] is False
    assert bench.parity_check([g("a", "evm", 1, "11"), g("a", "poseidon", 0, "11"), g("a", "evm", "survey:0", "11")], cpu)["bytes_equal"] is None
    out = dict(gpu_proofs=[g("a", "evm", 0, "11")], cpu_baseline=dict(proof_sha256="11"), configs=dict(x=dict(proof_sha256="5", cpu_baseline=dict(proof_sha256="6"))))
    bench.CPU_PROOFS[:] = cpu
    try:
        assert bench.finish_parity(out) is True and out["cpu_baseline"]["bytes_equal"] is True and out["configs"]["x"]["cpu_baseline"]["bytes_equal"] is False
    finally:
        bench.CPU_PROOFS[:] = []


def test_cpu_pass_records_the_oracle_proof_digest():
    import bench
    import halo2_zkcert_amd.prover as pv
    from oracle_backend import OracleBackend

    sh = pv.CircuitShape.agg(7, 3, 1)
    bench.CPU_PROOFS[:] = []
    try:
        bench.cpu_pass_seconds(pv, sh, "evm", 2, repeats=2, warm=False, witness_seed=1)
        p = pv.Prover(OracleBackend(2), sh, satisfiable=True)
        want = hashlib.sha256(bytes(p.prove(p.witness(1), transcript="evm")["proof"])).hexdigest()
        (rec,) = bench.CPU_PROOFS
        split = rec.pop("split_s")      # where the pass went: commitments, transforms, total (the remainder is what stays on the CPU at the host-pointer patch levels)
        assert rec == dict(shape=sh.name, k=7, transcript="evm", witness=1, proof_sha256=want, proof_bytes=rec["proof_bytes"])
        assert set(split) == {"msm", "fft", "total"} and 0 < split["msm"] < split["total"] and 0 < split["fft"] < split["total"]
    finally:
        bench.CPU_PROOFS[:] = []


def _supervised(tmp_path, digest):
    import halo2_zkcert_amd.prover as pv

    sh = pv.CircuitShape.agg(7, 3, 1)
    os.makedirs(tmp_path, exist_ok=True)
    proofs = [dict(shape=sh.name, k=7, transcript="evm", witness=0, proof_sha256=digest, where="stand-in")]
    env = dict(os.environ, ZKHIP_BENCH_WORKER_SCRIPT=os.path.join(ROOT, "tests", "fake_bench_worker.py"), FAKE_PLAN="{}", FAKE_COUNT_DIR=str(tmp_path),
               FAKE_K="7", FAKE_GPU_PROOFS=json.dumps(proofs))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "ZKHIP_BENCH_ROLE"):
        env.pop(k, None)
    detail = os.path.join(str(tmp_path), "detail.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--agg-k", "7", "--no-ladder", "--detail-out", detail], capture_output=True, text=True,
                       timeout=300, cwd=ROOT, env=env)
    lines = []
    for ln in r.stdout.splitlines():
        if ln.strip().startswith("{"):
            # the N > 1 supervisor's line: under the driver's 4 KB, a parity summary and the CPU leg's numbers in it, everything else in the file it names
            short = json.loads(ln)
            assert len(ln) < 4096 and short["detail"] == detail and set(short["parity"]) == {"bytes_equal", "n_compared"} and "gpu_proofs" not in short
            d = json.load(open(detail))
            assert short["parity"]["bytes_equal"] == d["parity"]["bytes_equal"] and short["cpu_baseline"]["bytes_equal"] == d["cpu_baseline"]["bytes_equal"]
            assert short["cpu_baseline"]["value"] == d["cpu_baseline"]["value"] and short["cpu_baseline"]["cores"] >= 1
            lines.append(d)
    return r, lines


def test_supervisor_compares_the_workers_digests_with_the_cpu_leg(tmp_path):
    """N > 1: rank 0's GPU-free supervisor times the CPU oracle once the workers are gone and compares its proof digest with the ones the
    workers' line carries; equal -> exit 0 and bytes_equal true; different -> the line still comes out and the run fails"""
    import halo2_zkcert_amd.prover as pv
    from oracle_backend import OracleBackend

    sh = pv.CircuitShape.agg(7, 3, 1)
    p = pv.Prover(OracleBackend(2), sh, satisfiable=True)
    good = hashlib.sha256(bytes(p.prove(p.witness(0), transcript="evm")["proof"])).hexdigest()
    r, lines = _supervised(tmp_path / "a", good)
    assert r.returncode == 0, r.stderr[-2000:]
    d = lines[0]
    assert d["parity"]["bytes_equal"] is True and d["cpu_baseline"]["bytes_equal"] is True and d["cpu_baseline"]["proof_sha256"] == good
    r, lines = _supervised(tmp_path / "b", "00" * 32)
    assert r.returncode != 0, r.stderr[-2000:]          # the supervisor exits 3; torch.distributed.run reports a failed child as 1
    assert lines[0]["parity"]["bytes_equal"] is False and "PARITY FAILURE" in r.stderr


def test_the_recorded_rsa_k17_digest_is_this_trees_oracle():
    """tests/golden/cpu_oracle_proof_digests.json was written on the GPU box's host cores; the k = 17 row is cheap enough to regenerate here:
    the oracle in THIS tree still produces it (the k = 19 / 20 / 22 rows came from the same process family: profiles/r05_cpu_parity.json)"""
    import halo2_zkcert_amd.prover as pv
    from oracle_backend import OracleBackend

    with open(os.path.join(ROOT, "tests", "golden", "cpu_oracle_proof_digests.json")) as f:
        want = json.load(f)["digests"]
    assert set(want) == {"agg_k22_a3+1/evm/witness0", "sha256_k19/poseidon/witness0", "rsa_k17/poseidon/witness0", "agg_k20_a3+1/evm/witness0"}
    p = pv.Prover(OracleBackend(os.cpu_count() or 8), pv.CircuitShape.rsa(17), satisfiable=True)
    got = hashlib.sha256(bytes(p.prove(p.witness(0), transcript="poseidon")["proof"])).hexdigest()
    assert got == want["rsa_k17/poseidon/witness0"]


@pytest.mark.slow
@pytest.mark.parametrize("k", [19, 20])
def test_the_recorded_full_size_digests_regenerate_from_the_oracle_alone(k):
    """opt-in (-m slow; k = 19: ~3 min, k = 20: ~1 min on 8 cores): tools/regen_cpu_digests.py's rows at k = 19 (SHA-shaped, Poseidon) and k = 20 (the
    headline shape, Keccak) come out of the CPU oracle in THIS tree as committed — the digests the full-size -m gpu tests hold the HIP path's proof
    bytes to (tests/test_gpu_prover.py) need no GPU box to be reproduced.  k = 22: `python tools/regen_cpu_digests.py --k 22` (12 min on 8 cores)."""
    import halo2_zkcert_amd.prover as pv

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import regen_cpu_digests as rg

    with open(rg.FIXTURE) as f:
        want = json.load(f)["digests"]
    (key, sh, kind), = rg.rows_for(pv, [k])
    assert rg.oracle_digest(pv, sh, kind, os.cpu_count() or 8) == want[key]


def test_chain_leaf_groups():
    """bench.py --chain: who proves which leaf (BASELINE configs[4], /root/reference/src/tests/x509_aggregation.rs:20-110): one leaf per rank below 6
    ranks; from 6 on the extra ranks join the two SHA-shaped leaves in groups whose sizes are powers of two; every rank is in at most one group"""
    import bench

    flat = {j: [j] for j in range(4)}
    assert bench.chain_leaf_groups(4) == flat and bench.chain_leaf_groups(5) == flat and bench.chain_leaf_groups(8, grouped=False) == flat
    assert bench.chain_leaf_groups(6) == {0: [0], 1: [1, 4], 2: [2], 3: [3, 5]}
    assert bench.chain_leaf_groups(8) == {0: [0], 1: [1, 4], 2: [2], 3: [3, 5]}          # ranks 6, 7: no leaf (a third member would stop the rows dividing)
    assert bench.chain_leaf_groups(10) == {0: [0], 1: [1, 4, 5, 6], 2: [2], 3: [3, 7, 8, 9]}
    for n in range(4, 20):
        g = bench.chain_leaf_groups(n)
        ranks = [r for rs in g.values() for r in rs]
        assert len(ranks) == len(set(ranks)) and all(0 <= r < n for r in ranks) and all(len(rs) & (len(rs) - 1) == 0 for rs in g.values())
        assert all(rs[0] == j for j, rs in g.items()) and len(g[1]) == len(g[3]) and len(g[0]) == len(g[2]) == 1
