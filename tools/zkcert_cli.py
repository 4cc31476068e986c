#!/usr/bin/env python3
"""CLI-shaped driver for the proving commands of the reference (argument names of /root/reference/src/bin/cli.rs:95-211),
hot path only: each command runs `create_proof` for a circuit of the command's shape on the GPU library and
consumes / produces the artefacts the reference's commands pass to each other, in the layouts of halo2_zkcert_amd/formats.py:

  * `<params-path>/kzg_bn254_<k>.srs`  read like `gen_srs(k)` (cli.rs:222,306) if it exists; otherwise a synthetic SRS (public trapdoor)
    is generated and kept as `kzg_bn254_<k>.synthetic.srs`, a name the reference never reads;
  * `--pk-path`  `read_pk` (cli.rs:312,335,362,455,509): if the file exists the proving key — fixed and permutation columns in Lagrange,
    coefficient and extended form, the l-polynomials — comes from it and NOTHING of the key is generated; otherwise the synthetic keygen
    runs and writes it (`gen_pk(.., Some(path))`, cli.rs:247,268,294,402);
  * `--witness-path`  the advice columns and instance values (an .npz; there is no certificate parsing / witness synthesis here: with a
    key file the witness must come from whoever made the key; without one it is generated, and written if the path is given);
  * the proof path receives what `gen_snark_shplonk(.., Some(path))` writes (cli.rs:320,343,369,462): a bincode `Snark` (protocol part
    empty: it belongs to the verifier side) holding the instances and the proof bytes; `gen-x509-agg-evm-proof` writes the raw proof
    (`gen_evm_proof_shplonk`, cli.rs:519);
  * `--snark-paths` (aggregation commands, cli.rs:478-483 `read_snark`): the leaf snarks are read and checked (instances, proof length);
  * `--break-points-path` (cli.rs:496-499): the aggregation circuit's advice column count is taken from the break points file
    (b break points = b + 1 columns) instead of --agg-advice.
Transcripts as in the reference: Poseidon for the gen_snark_shplonk commands, Keccak for gen-x509-agg-evm-proof.

    python tools/zkcert_cli.py prove-rsa --k 17 --pk-path build/rsa_1.pk --proof-path build/rsa_1.proof
    python tools/zkcert_cli.py prove-unoptimized-sha256 --k 19
    python tools/zkcert_cli.py gen-x509-agg-evm-proof --agg-k 22 --snark-paths build/rsa_1.proof build/rsa_2.proof
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

    def common(p, k_flag, k_default, proof_flag, proof_default, pk_flag, pk_default):
        p.add_argument(k_flag, type=int, default=k_default, dest="k", help="k parameter for circuit")
        p.add_argument("-p", "--params-path", default="./params", help="setup parameters path")
        p.add_argument(pk_flag, default=pk_default, dest="pk_path", help="proving key path (read if it exists, else written by the synthetic keygen)")
        p.add_argument(proof_flag, default=proof_default, dest="proof_path", help="output proof file")
        p.add_argument("--witness-path", default=None, help=".npz of advice columns / instance values (read with a key file, else written)")
        p.add_argument("--no-pk-file", action="store_true", help="neither read nor write the proving key file")
        p.add_argument("--repeat", type=int, default=1, help="prove this many times (the last proof is written)")

    common(sub.add_parser("prove-rsa"), "--k", 17, "--proof-path", "./build/rsa_1.proof", "--pk-path", "./build/rsa_1.pk")
    common(sub.add_parser("prove-unoptimized-sha256"), "--k", 19, "--proof-path", "./build/unoptimized_sha256_1.proof", "--pk-path", "./build/unoptimized_sha256_1.pk")
    common(sub.add_parser("prove-zkevm-sha256"), "--k", 11, "--proof-path", "./build/zkevm_sha256_1.proof", "--pk-path", "./build/zkevm_sha256_1.pk")
    for name, proof in (("gen-x509-agg-proof", "./build/x509_agg.proof"), ("gen-x509-agg-evm-proof", "./build/x509_agg_evm.proof")):
        p = sub.add_parser(name)
        common(p, "--agg-k", 22, "--agg-proof-path", proof, "--agg-pk-path", "./build/x509_agg.pk")
        p.add_argument("--snark-paths", nargs="*", default=[], help="leaf snarks (read_snark, cli.rs:478-483)")
        p.add_argument("--break-points-path", default="./build/x509_agg_break_points.json")
        p.add_argument("--agg-advice", type=int, default=3, help="basic advice columns when there is no break points file")
        p.add_argument("--agg-lookup-advice", type=int, default=1)
    args = ap.parse_args(argv)

    import numpy as np

    import halo2_zkcert_amd.ffi as ffi
    import halo2_zkcert_amd.formats as fm
    import halo2_zkcert_amd.prover as pv

    sha = args.cmd in ("prove-unoptimized-sha256", "prove-zkevm-sha256")
    agg = args.cmd.startswith("gen-x509-agg")
    info = {}
    if agg:
        n_adv = args.agg_advice
        if os.path.exists(args.break_points_path):
            bp = fm.read_break_points(args.break_points_path, k=args.k)
            n_adv = fm.advice_columns_from_break_points(bp)[0]
            info["break_points"] = dict(path=args.break_points_path, advice_columns=n_adv)
        shape = pv.CircuitShape.agg(args.k, n_adv, args.agg_lookup_advice)
        snarks = []
        for sp in args.snark_paths:
            s_ = fm.SnarkFile.read(sp)
            snarks.append(dict(path=sp, instances=[len(c) for c in s_.instances], proof_bytes=len(s_.proof)))
        if snarks:
            info["snarks"] = snarks
    else:
        shape = pv.CircuitShape.sha256(args.k) if sha else pv.CircuitShape.rsa(args.k)
    ctx = ffi.Context(0)
    backend = pv.GpuBackend(ctx, ffi)
    backend.params_file = os.path.join(args.params_path, f"kzg_bn254_{args.k}.srs")
    t0 = time.perf_counter()
    key_file = None
    if not args.no_pk_file and os.path.exists(args.pk_path):
        key_file = fm.ProvingKeyFile.read(args.pk_path, n_perm_columns=len(shape.perm_columns), n_selectors=0)
    prover = pv.Prover(backend, shape, satisfiable=True, key_file=key_file)      # keygen-like setup, or the key from the file
    if key_file is not None:
        if not args.witness_path or not os.path.exists(args.witness_path):
            raise SystemExit(f"{args.pk_path} exists: the witness must come with it (--witness-path); there is no witness synthesis here")
        wit = prover.load_witness(args.witness_path)
    else:
        wit = prover.witness(0)
        if args.witness_path:
            os.makedirs(os.path.dirname(args.witness_path) or ".", exist_ok=True)
            prover.save_witness(wit, args.witness_path)
        if not args.no_pk_file:
            fixed_c = [np.asarray(c[0], dtype=np.uint64) for c in backend.commit(prover.fixed_coeff, lagrange=False)]
            sigma_c = [np.asarray(c[0], dtype=np.uint64) for c in backend.commit(prover.sigma_coeff, lagrange=False)]
            os.makedirs(os.path.dirname(args.pk_path) or ".", exist_ok=True)
            fm.ProvingKeyFile.from_prover(prover, fixed_c, sigma_c).write(args.pk_path)
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
    if evm:
        with open(args.proof_path, "wb") as f:
            f.write(trace["proof"])
    else:
        inst = [[pv.from_mont_host(v) for v in col] for col in wit["instance_values"]]
        fm.SnarkFile(b"", inst, trace["proof"]).write(args.proof_path)
    print(json.dumps({"command": args.cmd, "circuit": shape.name, "k": args.k, "transcript": "evm-keccak" if evm else "poseidon",
                      "proof_bytes": len(trace["proof"]), "proof_path": args.proof_path, "proof_file": "raw proof" if evm else "bincode Snark (instances + proof)",
                      "params": backend.params_source, "proving_key": dict(source=prover.key_source, path=None if args.no_pk_file else args.pk_path),
                      "setup_s": round(t_setup, 3), "create_proof_s": [round(t, 6) for t in times], **info}))


if __name__ == "__main__":
    main()
