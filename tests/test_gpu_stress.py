"""Opt-in stress on the GPU (`-m "gpu and slow"` or ZK_RUN_SLOW=1; not part of the driver's `-m gpu` budget — tests/README.md): determinism of the
proof path while other processes take turns on the device, and the host-slice MSM entry point under random chunkings and sources.  Both were written in
round 6 after a full-suite run produced ONE SHA-shaped k = 19 chain leaf whose bytes differed from the single-GPU chain's (never reproduced: 12 000 proofs
under six-process contention and three reruns of that test were clean)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.slow]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_proofs_are_deterministic_under_six_process_contention():
    """six processes on one GPU (2 x SHA-shaped k = 19, 2 x RSA k = 17, 2 x aggregation-shaped k = 18, each with a second idle context as bench.py --chain
    holds): every proof of every process equals the digest its instance gave when proved alone (/root/reference/src/tests/x509_aggregation.rs:20-110 is the
    multi-proof flow this protects)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "contend_stress.py"), "--seconds", "40", "--second-context"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["mismatches"] == 0 and d["proofs"] > 500 and all("error" not in c for c in d["children"]), d


def test_host_slice_msm_under_random_chunkings():
    """zkhip_msm_g1 from pageable and pinned slices of 2^17..2^21 scalars, every chunk count, interleaved with device-resident work on the context: always the
    device-resident one-column MSM's point (best_multiexp at the `curves` patch level: /root/reference/src/helpers.rs:233)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "host_msm_stress.py"), "--seconds", "40"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "MISMATCH" not in r.stdout, r.stdout[-2000:] + r.stderr[-1000:]
    assert ", 0 mismatches" in r.stdout.strip().splitlines()[-1]


def test_no_kernel_depends_on_a_fresh_allocation_being_zero(tmp_path):
    """ZKHIP_POISON=1 fills every fresh device allocation of the library (and every ffi.Context.empty tensor) with 0xA5 bytes — what a GPU shared with other
    processes hands out, where a lone process mostly sees zeros: the single-GPU chain's five proof digests (2 x RSA k = 17, 2 x SHA-shaped k = 19, aggregation
    k = 18) are the unpoisoned run's.  (The whole single-process parity suite under the same variable: `bash tools/poison_check.sh`; round 6: 197 + 8 tests green.)"""
    digests = []
    for poison in ("0", "1"):
        path = str(tmp_path / f"chain_{poison}.json")
        env = dict(os.environ, ZKHIP_POISON=poison) if poison == "1" else {k_: v_ for k_, v_ in os.environ.items() if k_ != "ZKHIP_POISON"}
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--chain", "--steps", "1", "--warmup", "1", "--agg-k", "18", "--no-cpu-baseline", "--detail-out", path],
                           capture_output=True, text=True, env=env, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        digests.append(json.load(open(path))["proof_sha256"])
    assert len(digests[0]) == 5 and digests[0] == digests[1]


def test_proof_bytes_do_not_depend_on_the_relative_timing_of_a_proofs_streams(tmp_path):
    """options debug_delay_us / debug_delay_main_us: every side-stream / third-stream section of a proof held back 5 ms, then the main stream held back 5 ms behind every section it
    has issued — what contention for the device does to a proof's streams, made deterministic.  The single-GPU chain's five proof digests are the undisturbed run's: no consumer
    lacks its event dependency on an overlapped section, no writer overtakes a section that still reads (`bash tools/delay_check.sh` runs the parity files under it too)."""
    digests = []
    for env_extra in ({}, {"ZKHIP_DEBUG_DELAY_US": "5000"}, {"ZKHIP_DEBUG_DELAY_MAIN_US": "5000"}):
        path = str(tmp_path / f"chain_{len(digests)}.json")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--chain", "--steps", "1", "--warmup", "1", "--agg-k", "18", "--no-cpu-baseline", "--detail-out", path],
                           capture_output=True, text=True, env=dict(os.environ, **env_extra), timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        digests.append(json.load(open(path))["proof_sha256"])
    assert len(digests[0]) == 5 and digests[0] == digests[1] == digests[2]
