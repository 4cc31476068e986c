#!/usr/bin/env python3
"""Determinism of the single-GPU proof path UNDER CONTENTION: P processes share one GPU, each proves its own fixed instance over and over on an
unsharded context, and every proof's sha256 must equal the digest the same instance gave when it was proved ALONE.  (Round 6: a `bench.py
--chain --gpus 6 --no-leaf-groups` run on one device produced a SHA-shaped k = 19 leaf proof whose bytes differed from the single-GPU chain's —
once; kernel timing between the streams of a proof changes when other processes take turns on the device, so a missing dependency that regular
timing hides shows up here.)
    python tools/contend_stress.py --mix sha19,rsa17,sha19,rsa17,agg18,agg18 --seconds 60 [--env ZKHIP_LATE_OVERLAP=2]
Prints one JSON object: per child the proofs made and the mismatches seen."""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]


def shape_of(pv, name):
    if name == "sha19":
        return pv.CircuitShape.sha256(19, n_advice=32, n_fixed=12), "poseidon"
    if name.startswith("rsa"):
        return pv.CircuitShape.rsa(int(name[3:])), "poseidon"
    if name.startswith("agg"):
        return pv.CircuitShape.agg(int(name[3:]), 3, 1), "evm"
    if name.startswith("sha"):
        return pv.CircuitShape.sha256(int(name[3:]), n_advice=32, n_fixed=12), "poseidon"
    raise ValueError(name)


def child(args):
    import halo2_zkcert_amd.ffi as ffi
    import halo2_zkcert_amd.prover as pv

    ctx = ffi.Context(0)
    second = ffi.Context(0) if args.second_context else None      # bench.py --chain holds two contexts per process
    sh, kind = shape_of(pv, args.child)
    p = pv.Prover(pv.GpuBackend(ctx, ffi), sh, satisfiable=True)
    if second is not None:
        pv.GpuBackend(second, ffi)
    w = p.witness(args.witness)
    ctx.synchronize()
    first = hashlib.sha256(bytes(p.prove_native(w, transcript=kind)["proof"])).hexdigest()
    if args.expect is None:      # solo: report the digest
        again = hashlib.sha256(bytes(p.prove_native(w, transcript=kind)["proof"])).hexdigest()
        print(json.dumps({"shape": args.child, "digest": first, "repeat_equal": again == first}))
        return
    # wait for the start signal (a file the parent creates once every child has built its key), then prove until the deadline
    while not os.path.exists(args.go):
        time.sleep(0.01)
    t_end = time.time() + args.seconds
    made, bad, bad_at = 0, 0, []
    if first != args.expect:
        bad, bad_at = 1, [-1]
    while time.time() < t_end:
        d = hashlib.sha256(bytes(p.prove_native(w, transcript=kind)["proof"])).hexdigest()
        made += 1
        if d != args.expect:
            bad += 1
            if len(bad_at) < 8:
                bad_at.append(made)
    print(json.dumps({"shape": args.child, "proofs": made, "mismatches": bad, "at": bad_at}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mix", default="sha19,rsa17,sha19,rsa17,agg18,agg18")
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--env", action="append", default=[], help="NAME=VALUE for the children (library options: ZKHIP_LATE_OVERLAP=2 serialises a proof's streams, ...)")
    ap.add_argument("--second-context", action="store_true")
    ap.add_argument("--child", default=None)
    ap.add_argument("--expect", default=None)
    ap.add_argument("--witness", type=int, default=1)
    ap.add_argument("--go", default=None)
    args = ap.parse_args()
    if args.child:
        child(args)
        return
    env = dict(os.environ)
    for kv in args.env:
        k_, v_ = kv.split("=", 1)
        env[k_] = v_
    names = args.mix.split(",")
    me = [sys.executable, os.path.abspath(__file__)]
    extra = ["--second-context"] if args.second_context else []
    expect = {}
    for nm in sorted(set(names)):      # alone on the device (same library options)
        r = subprocess.run(me + ["--child", nm, "--witness", str(args.witness)] + extra, capture_output=True, text=True, env=env, timeout=600)
        if r.returncode != 0:
            raise SystemExit(f"solo {nm} failed: {r.stderr[-1500:]}")
        d = json.loads(r.stdout.strip().splitlines()[-1])
        assert d["repeat_equal"], d
        expect[nm] = d["digest"]
    go = f"/tmp/contend_go_{os.getpid()}"
    if os.path.exists(go):
        os.unlink(go)
    kids = [subprocess.Popen(me + ["--child", nm, "--expect", expect[nm], "--seconds", str(args.seconds), "--witness", str(args.witness), "--go", go] + extra,
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for nm in names]
    time.sleep(20.0 + 2.0 * len(names))      # key / SRS set-up of every child (the children poll for the file)
    open(go, "w").close()
    res = []
    for kproc in kids:
        out, err = kproc.communicate(timeout=args.seconds + 600)
        if kproc.returncode != 0:
            res.append({"error": err[-800:]})
        else:
            res.append(json.loads(out.strip().splitlines()[-1]))
    os.unlink(go)
    print(json.dumps({"mix": names, "env": args.env, "seconds": args.seconds, "second_context": args.second_context, "children": res,
                      "mismatches": sum(c.get("mismatches", 0) for c in res), "proofs": sum(c.get("proofs", 0) for c in res)}))


if __name__ == "__main__":
    main()
