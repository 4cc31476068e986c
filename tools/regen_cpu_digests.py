#!/usr/bin/env python3
"""tests/golden/cpu_oracle_proof_digests.json from the CPU oracle ALONE — no GPU box, no bench.py.

Each row is sha256(Prover(OracleBackend(threads), shape, satisfiable=True).prove(witness(0), transcript)["proof"]): the schedule is the product
package's Python mirror of upstream's create_proof (halo2-zkcert_amd/prover.py, with its Python transcripts) running on ORACLE arithmetic
(oracle/zkoracle.c through tests/oracle_backend.py) — the same instance (key, witness, blinding draws, transcript) the HIP path proves in the
full-size -m gpu tests and in bench.py, which compare their proof bytes with these digests.  What this fixture pins is therefore "prover.hip (C++)
== prover.py on oracle arithmetic", byte for byte; the schedule's independent check is the byte-driven verifier + pairing (oracle/pyref.py
verify_proof_bytes).  The reference's calls this stands for: /root/reference/src/helpers.rs:233,299 (gen_snark_shplonk), src/bin/cli.rs:519.

    python tools/regen_cpu_digests.py                 # k = 17, 19, 20: about 4 minutes on 8 cores; compares with the committed file
    python tools/regen_cpu_digests.py --k 17 19 20 22 # + the k = 22 headline: about 12 more minutes on 8 cores (2 on the GPU box's 16)
    python tools/regen_cpu_digests.py --write         # rewrite the fixture (rows not regenerated in this run are kept)
Exit code 1 if a regenerated digest differs from the committed one (without --write)."""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
FIXTURE = os.path.join(ROOT, "tests", "golden", "cpu_oracle_proof_digests.json")


def rows_for(pv, ks):
    """(fixture key, shape, transcript) for each requested size: the BASELINE configurations at their own sizes, the headline shape at 20 / 22"""
    table = {17: (pv.CircuitShape.rsa(17), "poseidon"), 19: (pv.CircuitShape.sha256(19, n_advice=32, n_fixed=12), "poseidon"),
             20: (pv.CircuitShape.agg(20, 3, 1), "evm"), 22: (pv.CircuitShape.agg(22, 3, 1), "evm")}
    out = []
    for k in ks:
        if k not in table:
            raise SystemExit(f"regen_cpu_digests: no fixture row at k = {k} (rows: {sorted(table)})")
        sh, kind = table[k]
        out.append((f"{sh.name}/{kind}/witness0", sh, kind))
    return out


def oracle_digest(pv, shape, kind, threads):
    from oracle_backend import OracleBackend

    p = pv.Prover(OracleBackend(threads), shape, satisfiable=True)
    return hashlib.sha256(bytes(p.prove(p.witness(0), transcript=kind)["proof"])).hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, nargs="+", default=[17, 19, 20])
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 8)
    ap.add_argument("--write", action="store_true")
    args = ap.parse_args()
    import halo2_zkcert_amd.prover as pv

    have = json.load(open(FIXTURE))
    bad = 0
    for key, sh, kind in rows_for(pv, args.k):
        t0 = time.time()
        d = oracle_digest(pv, sh, kind, args.threads)
        old = have["digests"].get(key)
        verdict = "new" if old is None else ("same" if old == d else "DIFFERS from the committed " + old)
        print(f"{key}: {d}  ({time.time() - t0:.0f} s, {args.threads} threads) {verdict}", flush=True)
        bad += old is not None and old != d
        have["digests"][key] = d
    if args.write:
        have["what"] = ("sha256 of the proof bytes of the BASELINE configurations at full size from the CPU oracle alone: "
                        "Prover(OracleBackend, shape, satisfiable=True).prove(witness(0), transcript) — halo2-zkcert_amd/prover.py's schedule (the Python mirror of "
                        "upstream's create_proof) on oracle arithmetic (oracle/zkoracle.c through tests/oracle_backend.py). Generator: python tools/regen_cpu_digests.py "
                        "--k 17 19 20 22 --write (k = 19: ~3 min, k = 22: ~12 min on 8 cores). First written in round 5 from bench.py's CPU leg on the GPU box's host "
                        "cores (profiles/r05_cpu_parity.json); regenerated GPU-free in round 6 with identical digests.")
        json.dump(have, open(FIXTURE, "w"), indent=1)
        print("wrote", os.path.relpath(FIXTURE, ROOT))
    elif bad:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
