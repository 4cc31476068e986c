#!/usr/bin/env python3
"""CLI-shaped driver for the proving commands of the reference (argument names of /root/reference/src/bin/cli.rs:95-211),
hot path only: each command runs `create_proof` for a synthetic circuit of the command's shape on the GPU
library — there is no witness generation from certificates here — reads `<params-path>/kzg_bn254_<k>.srs` like the reference's
`gen_srs` if that file exists, otherwise generates a synthetic SRS (public trapdoor) and keeps it as `kzg_bn254_<k>.synthetic.srs`,
a name the reference never reads; writes the proof bytes to the proof path and prints the timing as one JSON line.  Transcripts as
in the reference: Poseidon for the gen_snark_shplonk commands (cli.rs:320,343,369,462), Keccak for gen-x509-agg-evm-proof (cli.rs:519).

    python tools/zkcert_cli.py prove-rsa --k 17 --proof-path build/rsa_1.proof
    python tools/zkcert_cli.py prove-unoptimized-sha256 --k 19
    python tools/zkcert_cli.py gen-x509-agg-proof --agg-k 22
    python tools/zkcert_cli.py gen-x509-agg-evm-proof --agg-k 22        # Keccak EvmTranscript, 64-byte points
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="cmd", required=True)

    def common(p, k_flag, k_default, proof_flag, proof_default):
        p.add_argument(k_flag, type=int, default=k_default, dest="k", help="k parameter for circuit")
        p.add_argument("-p", "--params-path", default="./params", help="setup parameters path")
        p.add_argument(proof_flag, default=proof_default, dest="proof_path", help="output proof file")
        p.add_argument("--repeat", type=int, default=1, help="prove this many times (the last proof is written)")

    common(sub.add_parser("prove-rsa"), "--k", 17, "--proof-path", "./build/rsa_1.proof")
    common(sub.add_parser("prove-unoptimized-sha256"), "--k", 19, "--proof-path", "./build/unoptimized_sha256_1.proof")
    common(sub.add_parser("prove-zkevm-sha256"), "--k", 11, "--proof-path", "./build/zkevm_sha256_1.proof")
    common(sub.add_parser("gen-x509-agg-proof"), "--agg-k", 22, "--agg-proof-path", "./build/x509_agg.proof")
    common(sub.add_parser("gen-x509-agg-evm-proof"), "--agg-k", 22, "--agg-proof-path", "./build/x509_agg_evm.proof")
    args = ap.parse_args(argv)

    import halo2_zkcert_amd.ffi as ffi
    import halo2_zkcert_amd.prover as pv

    sha = args.cmd in ("prove-unoptimized-sha256", "prove-zkevm-sha256")
    shape = pv.CircuitShape.sha256(args.k) if sha else pv.CircuitShape.rsa(args.k)
    ctx = ffi.Context(0)
    backend = pv.GpuBackend(ctx, ffi)
    backend.params_file = os.path.join(args.params_path, f"kzg_bn254_{args.k}.srs")
    t0 = time.perf_counter()
    prover = pv.Prover(backend, shape, satisfiable=not sha)   # keygen-like setup: SRS, fixed / sigma polynomials and cosets
    wit = prover.witness(0)
    ctx.synchronize()
    t_setup = time.perf_counter() - t0
    evm = args.cmd == "gen-x509-agg-evm-proof"
    kind = "evm" if evm else "poseidon"
    times = []
    for _ in range(max(1, args.repeat)):
        ctx.synchronize()
        t0 = time.perf_counter()
        trace = prover.prove_native(wit, transcript=kind)
        times.append(time.perf_counter() - t0)
    os.makedirs(os.path.dirname(args.proof_path) or ".", exist_ok=True)
    with open(args.proof_path, "wb") as f:
        f.write(trace["proof"])
    print(json.dumps({"command": args.cmd, "circuit": shape.name, "k": args.k, "transcript": "evm-keccak" if evm else "poseidon",
                      "proof_bytes": len(trace["proof"]), "proof_path": args.proof_path, "params": backend.params_source,
                      "setup_s": round(t_setup, 3), "create_proof_s": [round(t, 6) for t in times]}))


if __name__ == "__main__":
    main()
