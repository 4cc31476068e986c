#!/usr/bin/env python3
"""bench.py — create_proof wall time on MI355X for the three configurations BASELINE.json's metric names.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = ONE zkhip_create_proof_ex call (halo2_proofs::plonk::create_proof from the point where the witness columns exist: every
commitment MSM, every NTT, the quotient sweep, lookup permute, grand products, evaluations and the SHPLONK multi-open, with a host
round trip at every Fiat-Shamir point) on a synthetic SATISFIABLE instance — the output is a proof the byte-driven verifier of the
test suite accepts.  Inputs (witness columns, SRS window tables, proving-key cosets) are resident in HBM before the timed region.

Headline (`value`): the aggregation-shaped k = 22 proof under the Keccak EvmTranscript (BASELINE configs[3], the configuration the
north-star target is stated on; /root/reference/src/bin/cli.rs:464-527).  `configs` carries all three configurations of the metric,
each timed with its own steps / warm-ups and its own rooflines: RSA k = 17 and zkevm-SHA256-shaped k = 19 under the Poseidon
transcript (what gen_snark_shplonk uses, cli.rs:320,369), aggregation k = 22 under Keccak.  --config picks another headline.

N > 1: the N processes torch.distributed.run starts (the driver's launch line, or `python bench.py --gpus N` on its own) are GPU-free
SUPERVISORS: each runs the real rank as a child process with a wall-clock budget, and together they walk a ladder of fresh worker sets
(row-sharded -> row-sharded on one communicator -> all-gather exchange -> MSMs by column -> independent proofs) until one rung completes on every rank; the line says which
rung it is (`ladder`, `comm_note`) and the run fails only if every rung does.  Inside the library every host wait of a multi-rank context
has a deadline (comm_timeout_ms), so a rank stuck in a collective dies loudly instead of spinning.
The workers (one process per GPU, RCCL): ONE k = 22 proof sharded over the ranks — every MSM by point range (window tables sharded 1/N),
coset NTTs by polynomial, the quotient sweep by row range; partial sums / columns / h exchanged with ncclAllGather inside the library
(zkhip_comm_*), everything else replicated -> "scaling": "strong".  --replicas runs N independent proofs instead ("weak");
--chain runs BASELINE configs[4] (2 x RSA + 2 x SHA leaf proofs on 4 ranks, barrier, then the sharded aggregation proof).
Prints ONE JSON line (rank 0).

Parity (round 5): every proof the CPU leg makes and every distinct proof of the HIP path is kept by sha256; the line's `parity` block lists the
pairs both legs proved (RSA k = 17 at its own size and the headline SHAPE at the CPU sample's size in a default run; the headline itself with
--cpu-baseline-k 22) and bench.py EXITS NON-ZERO when a pair differs — north_star's "proof bytes bit-identical to the CPU prover on the same SRS and
witness" is checked in every run, not only in tests.
--replay-rank R --of N: a MEASUREMENT MODE on one GPU — this process runs exactly what rank R of an N-rank proof runs (the library's RCCL branch bound
to tools/replay_rccl, which fabricates what the peers would send and can hold the communicator's stream for a modelled wire time), alone.  The line
is an N = 1 line with a `replay` block that says so; its proof bytes are wrong by construction and are offered for no comparison.
"""
import argparse
import glob
import hashlib
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s
MAD_PEAK_T = 29.8           # v_mad_u64_u32 lane-ops/s chip-wide, measured (profiles/r01_microbench_gfx950.txt): the integer roof
KERNELS = ("msm_digits", "msm_plan", "msm_accum_affine", "msm_accum_jac", "msm_tail", "ntt_strided", "ntt_final", "sweep",
           "lookup_permute", "grand_product", "batch_invert", "eval_polynomial", "linear_combination", "kate_division")


def host_threads():
    """Cores this process may actually use: scheduler affinity, capped by a cgroup CPU quota if one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = max(1, min(n, q // per))
        except Exception:
            pass
    return n


def build_hash():
    """Identifies the DEVICE code of the library: sha256 of libzkhip.so's .hip_fatbin section (every kernel of every translation unit, as
    compiled).  The PMC traffic figures under profiles/ are quoted only for the build they were measured on (`# build=<hash>` in their first
    line).  Host-only edits — schedule, transcripts, communicator, waits — leave the section byte-identical (checked: round 4), so they
    neither invalidate a counter pass nor have any reason to be shaped around one (until round 3 the key was a hash of six SOURCE files,
    host code included)."""
    import struct

    b = open(os.path.join(ROOT, "halo2-zkcert_amd", "libzkhip.so"), "rb").read()
    if b[:4] != b"\x7fELF" or b[4] != 2:
        raise RuntimeError("libzkhip.so is not a 64-bit ELF file")
    shoff = struct.unpack_from("<Q", b, 0x28)[0]
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", b, 0x3A)
    secs = [struct.unpack_from("<IIQQQQIIQQ", b, shoff + i * shentsize) for i in range(shnum)]
    names = secs[shstrndx][4]
    for sec in secs:
        if b[names + sec[0]: b.index(b"\0", names + sec[0])] == b".hip_fatbin":
            return hashlib.sha256(b[sec[4]: sec[4] + sec[5]]).hexdigest()[:16]
    raise RuntimeError("libzkhip.so has no .hip_fatbin section")


def make_shape(pv, name, args):
    if name == "rsa17":
        return pv.CircuitShape.rsa(17)
    if name == "sha19":
        return pv.CircuitShape.sha256(19, n_advice=args.sha_advice, n_fixed=args.sha_fixed)
    if name == "agg22":
        return pv.CircuitShape.agg(args.agg_k, args.agg_advice, args.agg_lookup_advice)
    raise ValueError(name)


TRANSCRIPT = {"rsa17": "poseidon", "sha19": "poseidon", "agg22": "evm"}
REFERENCE_CMD = {"rsa17": "prove-rsa (cli.rs:296-321, gen_snark_shplonk)", "sha19": "prove-zkevm-sha256 (cli.rs:345-370, gen_snark_shplonk)",
                 "agg22": "gen-x509-agg-evm-proof (cli.rs:464-527, gen_evm_proof_shplonk)"}


def pmc_traffic(config, bh):
    """HBM bytes per dispatch of the three hot kernels from the committed rocprofv3 --pmc passes of THIS build
    (profiles/*_pmc_<config>.csv, written by tools/profile_round.sh with a `# build=<hash>` header).  FETCH_SIZE / WRITE_SIZE are in
    KiB; FETCH_SIZE is doubled for the streaming kernels (NTT, sweep: 16 B per lane coalesced reads count at half their bytes on
    gfx950, MI355X_MICROARCH.md §HBM) and taken as is for the MSM accumulation's 64-byte table gathers (calibrated in round 1)."""
    import csv

    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_pmc_{config}.csv"))):
        with open(f) as fh:
            first = fh.readline()
            if not first.startswith("# build=") or first.strip().split("=")[1] != bh:
                continue
            rows = list(csv.DictReader(fh))
        best = (f, rows)
    if best is None:
        return None, None
    per, calls = {}, {}
    for r in best[1]:
        per[(r["counter"], r["kernel"])] = float(r["avg_per_dispatch"]) * 1024.0
        calls[r["kernel"]] = max(calls.get(r["kernel"], 0), int(float(r["dispatches"])))
    def tr(kernel, fetch_scale):
        f, w = per.get(("FETCH_SIZE", kernel)), per.get(("WRITE_SIZE", kernel))
        return None if f is None or w is None else f * fetch_scale + w
    def ntt_kernel(stem):
        """the register-tiled variant this configuration's transforms ran on (8 or 4 elements per thread, 2048 / 1024 tiles: ntt.hip picks by size)"""
        have = [k for k in calls if k.startswith(stem + "_r")]
        return max(have, key=lambda k: calls[k]) if have else stem + "_r4"
    ks, kf = ntt_kernel("k_ntt_strided"), ntt_kernel("k_ntt_final")
    out = {"msm_accum_affine": tr("k_accum_affine", 1.0), "ntt_strided": tr(ks, 2.0), "ntt_final": tr(kf, 2.0),
           "sweep": tr("k_sweep", 2.0), "ntt_kernels": f"{ks} + {kf}"}
    return out, os.path.relpath(best[0], ROOT)


CPU_PROOFS = []     # every proof the CPU leg made in this process: dict(shape, k, transcript, witness, proof_sha256, proof_bytes)


def cpu_pass_seconds(pv, shape, transcript, threads, repeats=3, warm=True, witness_seed=0):
    """median of `repeats` full passes of the same schedule on the CPU oracle (oracle/zkoracle.c, OpenMP), after one warm-up pass.
    The oracle proves the SAME instance as the GPU leg (same shape -> same key, witness(witness_seed), same blinding seeds, same transcript):
    the sha256 of its proof bytes is recorded in CPU_PROOFS so the line can say whether the GPU's bytes equal them (parity_check())."""
    sys.path[:0] = [p for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")) if p not in sys.path]
    from oracle_backend import OracleBackend

    class Timed(OracleBackend):
        """the same backend with a clock around the two things the `curves` / `domain` patch levels move to the GPU: best_multiexp (every commitment) and the transforms"""
        def __init__(self, threads_):
            super().__init__(threads_)
            self.clock_s = dict(msm=0.0, fft=0.0)

        def _clocked(self, key, f, *a):
            t_ = time.perf_counter()
            r = f(*a)
            self.clock_s[key] += time.perf_counter() - t_
            return r

        def partial_commit(self, *a):
            return self._clocked("msm", super().partial_commit, *a)

        def lagrange_to_coeff(self, *a):
            return self._clocked("fft", super().lagrange_to_coeff, *a)

        def coeff_to_lagrange(self, *a):
            return self._clocked("fft", super().coeff_to_lagrange, *a)

        def coeff_to_extended(self, *a):
            return self._clocked("fft", super().coeff_to_extended, *a)

        def divide_and_to_coeff(self, *a):
            return self._clocked("fft", super().divide_and_to_coeff, *a)

    backend = Timed(threads)
    p = pv.Prover(backend, shape, satisfiable=True)
    w = p.witness(witness_seed)
    if warm:
        p.prove(w, transcript=transcript)
    ts, digests, splits = [], set(), []
    for _ in range(repeats):
        backend.clock_s = dict(msm=0.0, fft=0.0)
        t0 = time.perf_counter()
        pf = bytes(p.prove(w, transcript=transcript)["proof"])
        ts.append(time.perf_counter() - t0)
        splits.append(dict(backend.clock_s, total=ts[-1]))
        digests.add(hashlib.sha256(pf).hexdigest())
    if len(digests) != 1:
        raise RuntimeError(f"the CPU oracle's proof of {shape.name} is not deterministic: {sorted(digests)}")
    med_split = sorted(splits, key=lambda d_: d_["total"])[len(splits) // 2]
    CPU_PROOFS.append(dict(shape=shape.name, k=shape.k, transcript=transcript, witness=witness_seed, proof_sha256=digests.pop(), proof_bytes=len(pf),
                           split_s={k_: round(v_, 4) for k_, v_ in med_split.items()}))
    return statistics.median(ts), ts


def parity_check(gpu_proofs, cpu_proofs=None):
    """north_star: "proof bytes bit-identical to the reference CPU prover on the same SRS and witness".  gpu_proofs: [dict(shape, k, transcript,
    witness, proof_sha256, ...)] made by the HIP path in this run; every one whose (shape, transcript, witness) the CPU oracle also proved is
    compared by digest.  -> dict(compared=[...], bytes_equal=True / False / None (nothing comparable))"""
    cpu = {(c["shape"], c["transcript"], c["witness"]): c for c in (CPU_PROOFS if cpu_proofs is None else cpu_proofs)}
    rows = []
    for g in gpu_proofs:
        c = cpu.get((g["shape"], g["transcript"], g["witness"]))
        if c is None:
            continue
        rows.append(dict(shape=g["shape"], k=g["k"], transcript=g["transcript"], witness=g["witness"], where=g.get("where"), proof_bytes=c["proof_bytes"],
                         gpu_sha256=g["proof_sha256"], cpu_sha256=c["proof_sha256"], equal=g["proof_sha256"] == c["proof_sha256"]))
    return dict(compared=rows, bytes_equal=(all(r["equal"] for r in rows) if rows else None),
                note="sha256 of the HIP path's proof bytes against the CPU oracle's (oracle/zkoracle.c through tests/oracle_backend.py) on the same "
                     "key, witness, blinding draws and transcript; a mismatch makes bench.py exit non-zero")


def recorded_cpu_k22():
    """profiles/r04_cpu_k22.json: ONE process on the GPU box's host cores that timed the CPU oracle on the headline shape at k = 18, 20 and 22
    (bench.py --cpu-baseline-k 22) -> dict or None"""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r04_cpu_k22.json")))["cpu_baseline"]
        return d if d.get("measured_k") == 22 else None
    except Exception:   # noqa: BLE001
        return None


def cpu_baseline(pv, args, config, head_k, transcript):
    """The CPU leg of the line: the same schedule on oracle/zkoracle.c (OpenMP, all usable host cores), SRS / keygen excluded.  GPU-free.
    The headline is k = 22: one CPU pass at that size takes two minutes, so the default run MEASURES the same circuit shape and transcript at
    k = 20 (one full pass) and k = 18 (median of 3 after a warm-up) — a bounded sample — and carries the k = 20 figure to k = 22 with the
    k = 20 -> 22 time ratio of the recorded real pass (profiles/r04_cpu_k22.json, same box kind, quoted beside it as measured_at_k22); without
    a record, with the measured k = 18 -> 20 growth.  --cpu-baseline-k 22 times real passes at all three sizes (scale 1).
    -> (dict for the line, dict for configs.rsa17 or None)"""
    threads = host_threads()

    def shape_at(k_):
        return make_shape(pv, config, argparse.Namespace(**{**vars(args), "agg_k": k_}))
    if config == "agg22" and head_k > 18:
        k_m = max(19, min(head_k, args.cpu_baseline_k))
        rec = recorded_cpu_k22() if head_k == 22 else None
        use_rec = bool(rec and k_m == 20 and rec.get("k20_s") and rec.get("k18_s") and rec.get("cores") == threads)
        t20 = None
        if use_rec:
            # bounded sample (10-30 s of CPU work): ONE pass at k = 20; the k = 18 figure and the growth are the record's
            med18, ts18 = rec["k18_s"], []
        else:
            med18, ts18 = cpu_pass_seconds(pv, shape_at(18), transcript, threads)
            if k_m > 20:
                t20, _ = cpu_pass_seconds(pv, shape_at(20), transcript, threads, repeats=1, warm=False)
        t_m, _ = cpu_pass_seconds(pv, shape_at(k_m), transcript, threads, repeats=1, warm=False)
        per4 = (t_m / med18) ** (2.0 / (k_m - 18))          # growth per 4x rows, k = 18 -> k_m
        if k_m == head_k:
            scale, how = 1.0, "a real pass at the headline size"
        elif use_rec:
            scale = rec["value"] / rec["k20_s"]
            how = (f"x{scale:.3f} = the k = 22 / k = 20 time ratio of the recorded real passes (profiles/r04_cpu_k22.json: {rec['value']:.1f} s / {rec['k20_s']:.1f} s / "
                   f"{rec['k18_s']:.2f} s at k = 22 / 20 / 18 on {rec['cores']} cores, one process)")
        else:
            scale = per4 ** ((head_k - k_m) / 2.0)
            how = f"x{scale:.3f} = the measured k = 18 -> {k_m} growth carried to k = {head_k}"
        k18_txt = (f"k = 18 in the record: {med18:.3f} s" if use_rec else
                   f"the same shape at k = 18: median of 3 after a warm-up = {med18:.3f} s ({', '.join(f'{t:.3f}' for t in ts18)})")
        out = dict(value=round(t_m * scale, 4), unit="s", cores=threads, kind="port", measured_s=round(t_m, 4), scale=round(scale, 4),
                   sample=f"{shape_at(k_m).name}: ONE full pass at k = {k_m} = {t_m:.3f} s; {k18_txt}; growth per 4x rows k = 18 -> {k_m} = {per4:.3f}; {how}"
                          "; oracle/zkoracle.c with OpenMP, SRS / keygen excluded",
                   sample_short=f"{shape_at(k_m).name}: one pass at k={k_m} = {t_m:.2f} s" + ("" if k_m == head_k else f" x{scale:.3f} ({'recorded k22/k20 ratio' if use_rec else 'measured growth'})") + f"; zkoracle.c, {threads} OpenMP threads",
                   measured_k=k_m, k18_s=round(med18, 4), growth_per_4x_rows=round(per4, 4),
                   # `value` is a MEASUREMENT only when the timed pass ran at the headline size; otherwise it is the measured pass carried to the
                   # headline size: say so in a field, not only in the sample text
                   extrapolated=(k_m != head_k), extrapolation=(None if k_m == head_k else ("recorded k = 22 / k = 20 ratio (profiles/r04_cpu_k22.json)" if use_rec else f"measured k = 18 -> {k_m} growth")))
        if t20 is not None:
            out["k20_s"] = round(t20, 4)
        if rec and k_m != head_k:
            out["measured_at_k22"] = dict(value=rec["value"], unit="s", cores=rec["cores"], k20_s=rec.get("k20_s"), k18_s=rec.get("k18_s"),
                                          source="profiles/r04_cpu_k22.json (one real pass of this shape at k = 22 on the GPU box's host cores, round 4)")
    else:
        sh = shape_at(head_k) if config == "agg22" else make_shape(pv, config, args)
        med, ts = cpu_pass_seconds(pv, sh, transcript, threads)
        out = dict(value=round(med, 4), unit="s", cores=threads, kind="port", measured_s=round(med, 4), scale=1.0, extrapolated=False, extrapolation=None, measured_k=sh.k,
                   sample_short=f"{sh.name}: median of 3 full passes = {med:.3f} s; zkoracle.c, {threads} OpenMP threads",
                   sample=f"{sh.name}: median of 3 full passes after a warm-up ({', '.join(f'{t:.3f}' for t in ts)}); "
                          "oracle/zkoracle.c with OpenMP, SRS / keygen excluded")
    out["proof_k"], out["proof_sha256"] = CPU_PROOFS[-1]["k"], CPU_PROOFS[-1]["proof_sha256"]     # the pass that was timed last = the measured size
    sp = CPU_PROOFS[-1].get("split_s")
    if sp and sp.get("total"):
        # where the CPU pass went: commitments (best_multiexp), transforms, and everything else — the part that stays on the CPU at the `curves` / `domain` patch levels
        rest = max(0.0, sp["total"] - sp["msm"] - sp["fft"])
        out["split_s"] = dict(msm=sp["msm"], fft=sp["fft"], rest=round(rest, 4), total=sp["total"], k=CPU_PROOFS[-1]["k"],
                              rest_fraction=round(rest / sp["total"], 4))
    rsa = None
    if config != "rsa17" and not args.no_other_configs and args.gpus == 1 and not args.chain:
        med17, ts17 = cpu_pass_seconds(pv, pv.CircuitShape.rsa(17), "poseidon", threads)
        rsa = dict(value=round(med17, 4), unit="s", cores=threads, kind="port", proof_k=17, proof_sha256=CPU_PROOFS[-1]["proof_sha256"],
                   sample=f"rsa_k17: median of 3 full passes after a warm-up ({', '.join(f'{t:.3f}' for t in ts17)})")
    return out, rsa


def chain_cpu_baseline(pv, args):
    """BASELINE configs[4] on the CPU oracle: the five proofs one after the other.  Bounded sample: the RSA k = 17 proof (median of 3), ONE
    pass of the SHA-shaped k = 19 proof, and the aggregation proof as cpu_baseline() samples it; value = 2 x rsa + 2 x sha + agg."""
    threads = host_threads()
    agg, _ = cpu_baseline(pv, args, "agg22", args.agg_k, "evm")
    med17, _ = cpu_pass_seconds(pv, pv.CircuitShape.rsa(17), "poseidon", threads)                                           # leaf 0's instance (witness 0)
    t19, _ = cpu_pass_seconds(pv, make_shape(pv, "sha19", args), "poseidon", threads, repeats=1, warm=False, witness_seed=1)   # leaf 1's instance (witness 1)
    return dict(value=round(2 * med17 + 2 * t19 + agg["value"], 4), unit="s", cores=threads, kind="port", extrapolated=agg.get("extrapolated"), measured_k=agg.get("measured_k"),
                sample_short=f"2 x rsa_k17 ({med17:.2f} s) + 2 x sha_k19 ({t19:.2f} s) + agg ({agg['value']:.1f} s{', extrapolated' if agg.get('extrapolated') else ''}); {threads} threads",
                sample=f"2 x rsa_k17 ({med17:.3f} s, median of 3) + 2 x sha256_k19 ({t19:.3f} s, one pass) + the aggregation proof ({agg['value']:.3f} s: {agg['sample']})",
                parts=dict(rsa17_s=round(med17, 4), sha19_s=round(t19, 4), agg=agg))


def finish_parity(out):
    """out["parity"] = every HIP-path proof of the line (out["gpu_proofs"]) against the CPU leg's digests; cpu_baseline.bytes_equal = the row of
    the CPU pass that was timed (the headline size with --cpu-baseline-k 22, else the parity sample's size)"""
    out["parity"] = parity_check(out.get("gpu_proofs") or [])
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict) and cb.get("proof_sha256"):
        rows = [r for r in out["parity"]["compared"] if r["cpu_sha256"] == cb["proof_sha256"]]
        cb["bytes_equal"] = all(r["equal"] for r in rows) if rows else None
    for name, c in (out.get("configs") or {}).items():
        cb_ = c.get("cpu_baseline") if isinstance(c, dict) else None
        if isinstance(cb_, dict) and cb_.get("proof_sha256") and c.get("proof_sha256"):
            cb_["bytes_equal"] = cb_["proof_sha256"] == c["proof_sha256"]
    return out["parity"]["bytes_equal"]


LINE_LIMIT = 4096      # the driver reads the last stdout line; round 5's 24 KB line came back unparsed (BENCH_r05.parsed == null)


def _cut(s, n):
    return s if not isinstance(s, str) or len(s) <= n else s[: n - 3] + "..."


def compact_line(out, detail_path):
    """The ONE stdout line the driver parses: the contract's fields, `roofline` (incl. traffic), `int_roofline` (numbers), `cpu_baseline` (numbers,
    `extrapolated`, `bytes_equal`), `parity` {bytes_equal, n_compared} and the name of the detail file; every string short.  Everything else of `out`
    (configs, gpu_proofs, chain, comm, replay, the long notes) lives in the detail file only."""
    cfg = out.get("config") or {}
    line = {k_: out.get(k_) for k_ in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline")}
    line["dtype"] = "u256"
    line["data"] = out.get("data", "synthetic")
    line["proofs_per_step"] = out.get("proofs_per_step")
    line["config"] = {"workload": _cut(cfg.get("workload_short") or cfg.get("workload"), 126)}
    for k_ in ("headline", "k", "advice", "fixed", "lookups", "perm_columns", "degree", "transcript"):
        if k_ in cfg:
            line["config"][k_] = cfg[k_]
    line["config"]["parallelism"] = _cut(cfg.get("parallelism_short") or cfg.get("parallelism"), 126)
    rf = out.get("roofline")
    if isinstance(rf, dict):
        line["roofline"] = {k_: rf.get(k_) for k_ in ("kernel", "bound", "achieved", "peak", "unit", "frac", "hbm_frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms") if k_ in rf}
    else:
        line["roofline"] = None
    ir = out.get("int_roofline")
    if isinstance(ir, dict):
        line["int_roofline"] = {k_: ir.get(k_) for k_ in ("kernel", "achieved", "peak", "unit", "frac")}
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        c = {k_: cb[k_] for k_ in ("value", "unit", "cores", "kind", "measured_s", "measured_k", "scale", "extrapolated", "bytes_equal", "error") if k_ in cb}
        if "sample" in cb:
            c["sample"] = _cut(cb.get("sample_short") or cb["sample"], 126)
        if isinstance(cb.get("measured_at_k22"), dict):
            c["measured_at_k22_s"] = cb["measured_at_k22"].get("value")
        line["cpu_baseline"] = c
    else:
        line["cpu_baseline"] = None
    par = out.get("parity")
    line["parity"] = {"bytes_equal": par.get("bytes_equal"), "n_compared": len(par.get("compared") or [])} if isinstance(par, dict) else None
    for k_ in ("build", "setup_s", "first_proof_s"):
        if out.get(k_) is not None:
            line[k_] = out[k_]
    if isinstance(out.get("with_h2d"), dict):
        line["with_h2d_s"] = out["with_h2d"].get("value")
    if isinstance(out.get("ffi_levels"), dict):
        line["ffi_levels_s"] = {k_: out["ffi_levels"][k_].get("value") for k_ in ("curves", "domain", "one-call") if isinstance(out["ffi_levels"].get(k_), dict)}
    if isinstance(out.get("configs"), dict):
        line["configs_s"] = {k_: v_.get("value") for k_, v_ in out["configs"].items() if isinstance(v_, dict) and "value" in v_}
    if isinstance(out.get("ladder"), dict):
        line["ladder"] = {"rung": out["ladder"].get("rung"), "of": out["ladder"].get("of"), "label": _cut(out["ladder"].get("label"), 100)}
    if isinstance(out.get("comm"), dict):
        line["comm"] = {k_: out["comm"].get(k_) for k_ in ("transport", "nranks", "transport_ranks", "shard_mode", "collectives_per_step", "bytes_gathered_per_step")}
    if isinstance(out.get("replay"), dict):
        line["replay"] = {"rank": out["replay"].get("rank"), "of": out["replay"].get("of"), "note": "single-rank replay on 1 GPU, not an N-GPU measurement"}
    line["detail"] = detail_path
    text = json.dumps(line)
    for victim in ("configs_s", "ffi_levels_s", "comm", "ladder", "replay"):     # never needed with the strings bounded as above; a guarantee, not a plan
        if len(text) < LINE_LIMIT:
            break
        line.pop(victim, None)
        text = json.dumps(line)
    if len(text) >= LINE_LIMIT:
        raise RuntimeError(f"bench.py: the stdout line is {len(text)} bytes (limit {LINE_LIMIT})")
    return text


def emit(out, args):
    """Rank 0's last word: the whole result object goes to the detail file (--detail-out, default bench_detail.json next to this script), the compact
    line (compact_line) to stdout.  A worker rank of an N > 1 run prints the WHOLE object instead: its stdout is a temp file its supervisor reads."""
    if os.environ.get("ZKHIP_BENCH_ROLE") == "worker":
        print(json.dumps(out), flush=True)
        return
    path = getattr(args, "detail_out", None) or os.path.join(ROOT, "bench_detail.json")
    try:
        with open(path, "w") as fh:
            json.dump(out, fh, indent=1)
            fh.write("\n")
    except OSError:
        import tempfile

        with tempfile.NamedTemporaryFile("w", prefix="bench_detail_", suffix=".json", delete=False) as fh:
            json.dump(out, fh, indent=1)
            path = fh.name
    shown = os.path.relpath(path, ROOT) if os.path.abspath(path).startswith(ROOT + os.sep) else path
    print(f"bench.py: detail ({os.path.getsize(path)} bytes) in {shown}", file=sys.stderr, flush=True)
    print(compact_line(out, shown), flush=True)


def lib_transport():
    """What carries the library's own exchanges in an N > 1 worker: RCCL — whatever backend torch's control plane uses (the ladder's later rungs put torch on gloo so
    that the library's communicator is the only RCCL communicator of the process; ffi.Context.comm_init on its own would take a gloo backend to mean the host-staged
    test transport).  The host-staged transport only where RCCL cannot run: every rank on ONE device (ZKHIP_BENCH_ONE_DEVICE=1, the one-GPU tests), or when the
    environment names it (ZKHIP_COMM_TRANSPORT)."""
    return os.environ.get("ZKHIP_COMM_TRANSPORT") or ("host" if os.environ.get("ZKHIP_BENCH_ONE_DEVICE") == "1" else "rccl")


def chain_leaf_groups(n_ranks, grouped=True):
    """--chain: {leaf index: [ranks]} — who proves which of the four leaves (0, 2: RSA-shaped k = 17; 1, 3: SHA-shaped k = 19).  Leaf j's head is
    rank j.  From 6 ranks on (grouped) the ranks 4.. — idle until the aggregation proof otherwise — are dealt to the two SHA leaves, whose proofs take
    4.5x as long as the RSA ones.  Group sizes are powers of two (the sweep and the exchanges divide the rows by the group size; single-rank replay,
    k = 19: 31.0 ms on one rank, 18.6 over two, 19.4 over three): each SHA leaf takes 2^e - 1 of the extra ranks, the largest e that both can have —
    one each at 6-9 ranks, three each from 10."""
    groups = {j: [j] for j in range(4)}
    if grouped and n_ranks >= 6:
        extra = 1
        while 2 * (2 * extra + 1) <= n_ranks - 4:
            extra = 2 * extra + 1
        pool = list(range(4, n_ranks))
        for j in (1, 3):
            groups[j] += pool[:extra]
            pool = pool[extra:]
    return groups


LADDER = {
    # one sharded proof: each rung is a FRESH set of worker processes (a rank that touched the GPU is never reused or exec'ed over).
    # From the second rung on the library's bulk communicator is off too (ZKHIP_COMM_BULK=0: no ncclCommSplit, one communicator as in round 4):
    # whatever made the first rung fail, the fall-backs do not repeat its newest moving part.  The second rung keeps the row-sharded exchange (the
    # faster one by replay) and drops only what has never met more than one real GPU together with it: the second communicator on
    # every rank and torch's own RCCL communicator beside the library's
    "shard": [("row-sharded (all-to-all of row windows, torch control plane on RCCL)", [], {}),
              ("row-sharded on ONE communicator (no bulk communicator), torch control plane on gloo", [], {"ZKHIP_COMM_BULK": "0", "ZKHIP_BENCH_DIST_BACKEND": "gloo"}),
              ("all-gather exchange (row_sharded = 0), torch control plane on gloo", [], {"ZKHIP_ROW_SHARDED": "0", "ZKHIP_COMM_BULK": "0", "ZKHIP_BENCH_DIST_BACKEND": "gloo"}),
              ("MSMs by column, whole tables on every rank", ["--shard", "columns"], {"ZKHIP_ROW_SHARDED": "0", "ZKHIP_COMM_BULK": "0", "ZKHIP_BENCH_DIST_BACKEND": "gloo"}),
              ("independent proofs, one per GPU (no collective on the data path)", ["--replicas"], {"ZKHIP_BENCH_DIST_BACKEND": "gloo"})],
    "replicas": [("independent proofs, one per GPU", [], {}),
                 ("independent proofs, one per GPU, torch control plane on gloo", [], {"ZKHIP_BENCH_DIST_BACKEND": "gloo"})],
    "chain": [("one leaf per rank 0-3 (--leaf-groups: the SHA leaves over rank groups), aggregation proof row-sharded over all ranks", [], {}),
              ("one leaf per rank 0-3, aggregation proof row-sharded on ONE communicator, torch control plane on gloo", ["--no-leaf-groups"], {"ZKHIP_COMM_BULK": "0", "ZKHIP_BENCH_DIST_BACKEND": "gloo"}),
              ("one leaf per rank 0-3, aggregation proof with the all-gather exchange, torch control plane on gloo", ["--no-leaf-groups"], {"ZKHIP_ROW_SHARDED": "0", "ZKHIP_COMM_BULK": "0", "ZKHIP_BENCH_DIST_BACKEND": "gloo"}),
              ("one leaf per rank 0-3, aggregation proof with MSMs by column", ["--no-leaf-groups", "--shard", "columns"], {"ZKHIP_ROW_SHARDED": "0", "ZKHIP_COMM_BULK": "0", "ZKHIP_BENCH_DIST_BACKEND": "gloo"}),
              ("one leaf per rank 0-3, aggregation proof on rank 0 alone (no collective on the data path)", ["--no-leaf-groups", "--agg-unsharded"], {"ZKHIP_BENCH_DIST_BACKEND": "gloo"})],
}


def _free_port():
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _kill_group(child, grace=5.0):
    """SIGTERM, then SIGKILL, to the process GROUP the child leads (start_new_session=True): exactly what this process started"""
    import signal

    if child.poll() is not None:
        return
    for sig, wait in ((signal.SIGTERM, grace), (signal.SIGKILL, 10.0)):
        try:
            os.killpg(child.pid, sig)
        except ProcessLookupError:
            return
        t_end = time.monotonic() + wait
        while time.monotonic() < t_end:
            if child.poll() is not None:
                return
            time.sleep(0.1)


def self_launch(n, budget_s):
    """python -m torch.distributed.run --nnodes=1 --nproc-per-node n --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>
    as a child process in its own process group; -> its exit code.  The n processes it starts are SUPERVISORS (supervise() below): they never
    touch the GPU and run the fallback ladder of fresh worker processes.  This parent only holds the outermost deadline."""
    import subprocess

    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_threads() // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: launching " + " ".join(cmd), file=sys.stderr, flush=True)
    child = subprocess.Popen(cmd, env=env, start_new_session=True)
    try:
        return child.wait(timeout=budget_s)
    except subprocess.TimeoutExpired:
        print(f"bench.py: the {n}-rank run did not finish within {budget_s:.0f} s: killing it", file=sys.stderr, flush=True)
        _kill_group(child)
        return 124
    except BaseException:
        _kill_group(child)
        raise


def supervise(args):
    """One of the N processes torch.distributed.run started (by the driver, or by self_launch).  It NEVER touches the GPU: it runs the actual
    bench rank as a child process (ZKHIP_BENCH_ROLE=worker, own process group), with a wall-clock budget, and coordinates with the
    other supervisors through a TCP store.  A rung fails when any rank's worker exits non-zero or overruns its budget; every supervisor
    then kills its worker's process group and all of them start a FRESH worker on the next rung of LADDER (row-sharded -> row-sharded on one communicator -> all-gather
    exchange -> MSMs by column -> independent proofs).  Rank 0 relays the JSON line of the first rung that completes on every rank, adds
    the ladder's history and — workers gone, GPU idle — times the CPU baseline.  -> exit code (non-zero only if every rung failed)."""
    import datetime
    import signal
    import subprocess
    import tempfile

    from torch.distributed import TCPStore   # CPU only: no HIP call is made in this process

    rank, world, local_rank = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    addr, port = os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ["MASTER_PORT"])
    agent_store = os.environ.get("TORCHELASTIC_USE_AGENT_STORE") == "True"     # torch.distributed.run hosts the store at MASTER_PORT itself
    store = TCPStore(addr, port, world, is_master=(rank == 0 and not agent_store), timeout=datetime.timedelta(seconds=300), wait_for_workers=False)
    kind = "chain" if args.chain else ("replicas" if args.replicas else "shard")
    rungs = LADDER[kind] if not args.no_ladder else LADDER[kind][:1]
    t_begin = time.monotonic()
    history, current = [], {"child": None}

    def on_term(signum, frame):   # the launcher is tearing the job down: take the worker along
        if current["child"] is not None:
            _kill_group(current["child"], grace=2.0)
        os._exit(143)
    signal.signal(signal.SIGTERM, on_term)

    def count(key):
        return store.add(key, 0)

    for i, (label, extra_args, extra_env) in enumerate(rungs):
        key = f"zkbench/rung{i}"
        if rank == 0:
            store.set(key + "/port", str(_free_port()))
        wport = int(store.get(key + "/port"))
        later = len(rungs) - 1 - i
        budget = max(min(60.0, args.rung_budget), min(args.rung_budget, args.ladder_budget - (time.monotonic() - t_begin) - 150.0 * later))
        env = dict(os.environ, ZKHIP_BENCH_ROLE="worker", ZKHIP_BENCH_RUNG=label, MASTER_PORT=str(wport), **extra_env)
        env.setdefault("ZKHIP_COMM_TIMEOUT_MS", str(args.comm_timeout_ms))
        for k_ in ("TORCHELASTIC_USE_AGENT_STORE",):     # the workers rendezvous among themselves on their own port
            env.pop(k_, None)
        out_f = tempfile.NamedTemporaryFile(prefix=f"zkbench_r{rank}_", suffix=".out", delete=False)
        # (ZKHIP_BENCH_WORKER_SCRIPT: tests/fake_bench_worker.py stands in for the GPU rank in the CPU-only tests of this ladder)
        cmd = [sys.executable, os.environ.get("ZKHIP_BENCH_WORKER_SCRIPT") or os.path.abspath(__file__)] + sys.argv[1:] + extra_args
        if rank == 0:
            print(f"bench.py: rung {i + 1}/{len(rungs)} [{label}] budget {budget:.0f} s", file=sys.stderr, flush=True)
        child = subprocess.Popen(cmd, env=env, stdout=out_f, start_new_session=True)
        current["child"] = child
        t0, why = time.monotonic(), ""
        while True:
            rc = child.poll()
            if rc is not None:
                if rc != 0:
                    why = f"rank {rank}: worker exited with code {rc}"
                break
            if time.monotonic() - t0 > budget:
                why = f"rank {rank}: worker overran its {budget:.0f} s budget (killed)"
                _kill_group(child)
                break
            if count(key + "/fail") > 0:      # a peer's worker failed: mine cannot complete a collective run either
                time.sleep(3.0)               # (let it report its own error first)
                if child.poll() is None:
                    _kill_group(child)
                why = why or f"rank {rank}: stopped because a peer's worker failed"
                break
            time.sleep(0.25)
        current["child"] = None
        if why and child.returncode != 0:
            store.add(key + "/fail", 1)
            store.set(key + f"/why{rank}", why)
        store.add(key + "/done", 1)
        t_w = time.monotonic()
        while count(key + "/done") < world and time.monotonic() - t_w < budget + 60.0:
            time.sleep(0.1)
        failed = count(key + "/fail") > 0 or count(key + "/done") < world
        out_f.close()
        text = open(out_f.name, "r", errors="replace").read()
        os.unlink(out_f.name)
        if not failed:
            if rank == 0:
                lines = [ln for ln in text.splitlines() if ln.strip().startswith("{")]
                if not lines:
                    print("bench.py: the workers finished without a JSON line", file=sys.stderr, flush=True)
                    return 1
                out = json.loads(lines[-1])
                out["ladder"] = {"rung": i + 1, "of": len(rungs), "label": label, "failed_rungs": history,
                                 "note": "GPU-free supervisors (one per rank) run each rung in fresh worker processes with a wall-clock budget"}
                if history:
                    out["comm_note"] = "; ".join(f"rung {h['rung']} [{h['label']}] failed: {h['why']}" for h in history) + f"; this line is rung {i + 1} [{label}]"
                if out.get("cpu_baseline") is None and not args.no_cpu_baseline:
                    try:
                        import halo2_zkcert_amd.prover as pv
                        out["cpu_baseline"] = chain_cpu_baseline(pv, args) if args.chain else cpu_baseline(pv, args, args.config, out["config"].get("k", args.agg_k), out["config"].get("transcript", TRANSCRIPT[args.config]))[0]
                        finish_parity(out)
                    except Exception as e:   # noqa: BLE001 — the GPU measurement stands on its own
                        out["cpu_baseline"] = dict(error=str(e)[:300])
                emit(out, args)
                if isinstance(out.get("parity"), dict) and out["parity"].get("bytes_equal") is False:
                    print("bench.py: PARITY FAILURE: a HIP-path proof differs from the CPU oracle's: " + json.dumps([r for r in out["parity"]["compared"] if not r["equal"]]), file=sys.stderr, flush=True)
                    return 3
            return 0
        whys = []
        if rank == 0:
            for r in range(world):
                try:
                    if store.check([key + f"/why{r}"]):
                        whys.append(store.get(key + f"/why{r}").decode())
                except Exception:   # noqa: BLE001
                    pass
            print(f"bench.py: rung {i + 1} [{label}] FAILED: {'; '.join(whys) or 'no rank reported why'}", file=sys.stderr, flush=True)
        history.append({"rung": i + 1, "label": label, "why": "; ".join(whys)[:400] if whys else "see stderr"})
        time.sleep(1.0)     # let the killed workers' GPU queues be torn down before the next rung starts
    if rank == 0:
        print("bench.py: every rung of the ladder failed", file=sys.stderr, flush=True)
    return 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="agg22", choices=["agg22", "rsa17", "sha19"], help="the headline configuration (`value`)")
    ap.add_argument("--agg-k", type=int, default=22)
    ap.add_argument("--agg-advice", type=int, default=3, help="basic advice columns of the aggregation-shaped circuit (the real count is "
                    "what calculate_params(Some(10)) returns for 4 verified snarks, cli.rs:493 — unknown here, so a parameter)")
    ap.add_argument("--agg-lookup-advice", type=int, default=1)
    ap.add_argument("--sha-advice", type=int, default=32)
    ap.add_argument("--sha-fixed", type=int, default=12)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-h2d", action="store_true", help="skip the with_h2d passes (profiling runs: they split the advice commitment into two launches, "
                    "which changes the per-launch averages a kernel trace reports)")
    ap.add_argument("--no-other-configs", action="store_true", help="time the headline configuration only")
    ap.add_argument("--other-steps", type=int, default=10)
    ap.add_argument("--shard", default="auto", choices=["auto", "points", "columns"],
                    help="N > 1: how the MSMs of the sharded proof are split — by point range with 1/N of the window tables per rank (auto: "
                         "k >= 20), or by column with whole tables on every rank (auto: k <= 19, where one MSM cannot fill several GPUs)")
    ap.add_argument("--replicas", action="store_true", help="N > 1: N independent proofs, one per GPU (weak scaling) instead of one sharded proof")
    ap.add_argument("--chain", action="store_true", help="N >= 4: BASELINE configs[4] — 2 x RSA + 2 x SHA leaf proofs on 4 ranks, then the sharded aggregation proof")
    ap.add_argument("--python-schedule", action="store_true", help="drive the proof from prover.py over the small entry points (same proof bytes)")
    ap.add_argument("--no-chain", action="store_true", help="skip the configs.chain entry (BASELINE configs[4] on this GPU) of the default line")
    ap.add_argument("--cpu-baseline-k", type=int, default=20, help="rows (2^k) of the CPU oracle's timed pass for the k = 22 headline; the k = 18 pass "
                    "is timed beside it and the measured k-2 -> k ratio is what extrapolates (22 = one real pass, no extrapolation: minutes)")
    ap.add_argument("--rung-budget", type=float, default=420.0, help="N > 1: wall-clock seconds one rung of the fallback ladder may take before its workers are killed")
    ap.add_argument("--ladder-budget", type=float, default=1500.0, help="N > 1: wall-clock seconds for the whole ladder (the driver's own limit is 1800 s)")
    ap.add_argument("--no-ladder", action="store_true", help="N > 1: first rung only (a failure is the run's failure)")
    ap.add_argument("--comm-timeout-ms", type=int, default=60000, help="N > 1: the library's deadline for a host wait on a multi-rank context (zkhip_set_option comm_timeout_ms)")
    ap.add_argument("--replay-rank", type=int, default=None, help="MEASUREMENT MODE on one GPU: run what rank R of an --of N proof runs, alone, with the peers' "
                    "contributions fabricated on the device (tools/replay_rccl): rank R's kernels, launch structure and exchanges on an idle GPU. "
                    "The proof bytes are wrong by construction; the line reports rank R's time and what does not divide by N, never an N-GPU result")
    ap.add_argument("--of", type=int, default=0, help="with --replay-rank: the rank count N being replayed")
    ap.add_argument("--replay-latency-us", type=float, default=0.0, help="with --replay-rank: modelled latency of one exchange (0: exchanges cost only the fabricating fill)")
    ap.add_argument("--replay-link-gbs", type=float, default=0.0, help="with --replay-rank: modelled xGMI bandwidth per peer link and direction, GB/s (0: not modelled)")
    ap.add_argument("--msm-c", type=int, default=None, help="window width of the SRS tables built in this run (the library's msm_c option; default: by size)")
    ap.add_argument("--row-sharded", type=int, default=None, choices=[0, 1], help="sharded proofs: 0 = all-gathers of complete columns instead of the all-to-all of row windows "
                    "(the library's row_sharded option; default: the library's, 1)")
    ap.add_argument("--leaf-groups", action="store_true", help="--chain, N >= 6: the SHA-shaped leaves over rank GROUPS (ranks 4.. join them) instead of one rank per leaf.  Off by "
                    "default since round 6: replayed with a modelled wire the two-rank SHA leaf is slower than the one-rank leaf unless a link sustains ~100 GB/s "
                    "(profiles/r06_rank_replay.json: 20.7 ms with a free interconnect, 27.8 / 41.2 / 66.0 ms at 100 / 50 / 25 GB/s, against 32 ms alone)")
    ap.add_argument("--no-leaf-groups", action="store_true", help="--chain: every leaf proof on one rank (the default; kept for the ladder's rungs and older command lines)")
    ap.add_argument("--ffi-level", default="all", choices=["all", "curves", "domain", "one-call", "none"],
                    help="N = 1: the boundary timed at each patch level of INTEGRATION.md (`ffi_levels` in the detail file): `curves` = one proof's worth of best_multiexp / "
                         "best_fft calls through the HOST-pointer zkhip_msm_g1 / zkhip_fft with pageable arrays (what patching only halo2curves buys), `domain` = the same with "
                         "EvaluationDomain's host forms (zkhip_lagrange_to_coeff / coeff_to_extended / extended_to_coeff), `one-call` = the headline (zkhip_create_proof_ex)")
    ap.add_argument("--detail-out", default=None, help="where rank 0 writes the whole result object (default: bench_detail.json next to this script); "
                    "stdout carries one compact line (< 4 KB) that names it")
    ap.add_argument("--agg-unsharded", action="store_true", help="--chain, N >= 4: the aggregation proof on rank 0 alone (last rung of the ladder: no collective)")
    args = ap.parse_args()

    if args.replay_rank is not None:
        if args.gpus != 1 or args.of < 2 or not 0 <= args.replay_rank < args.of or args.replicas:
            raise SystemExit("bench.py: --replay-rank R needs --gpus 1 and --of N with 0 <= R < N, N >= 2")
        args.no_cpu_baseline = True      # a replayed rank's proof is wrong by construction: nothing to compare, nothing to time beside it
        args.no_other_configs = True
    if args.gpus > 1 and os.environ.get("ZKHIP_BENCH_ROLE") != "worker":
        # Neither of these two processes ever touches the GPU (no HIP call, no torch.cuda): children, never an exec.
        if "WORLD_SIZE" not in os.environ:
            # `python bench.py --gpus N` on its own: start N supervisors under torch.distributed.run and hold the outermost deadline
            sys.exit(self_launch(args.gpus, args.ladder_budget + 240.0))
        # one of the N processes torch.distributed.run started (the driver's launch line, or self_launch): the GPU-free supervisor of this rank
        sys.exit(supervise(args))
    if args.gpus > 1:
        # a worker: any failure ends the process at once, without the teardown of a communicator that may be stuck (the supervisor starts the next rung)
        try:
            worker(args)
            sys.stdout.flush()
            sys.stderr.flush()
        except BaseException as e:   # noqa: BLE001
            import traceback

            traceback.print_exc()
            print(f"bench.py worker rank {os.environ.get('RANK')}: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
            os._exit(e.code if isinstance(e, SystemExit) and isinstance(e.code, int) else 1)
        os._exit(0)
    worker(args)


def worker(args):
    import torch

    import halo2_zkcert_amd.ffi as ffi
    import halo2_zkcert_amd.prover as pv

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist

        # test hooks (a 1-GPU box can still run the N > 1 control flow): ZKHIP_BENCH_ONE_DEVICE=1 puts every rank on device 0,
        # ZKHIP_BENCH_DIST_BACKEND=gloo replaces RCCL for torch's own collectives (RCCL refuses two ranks on one device)
        if os.environ.get("ZKHIP_BENCH_ONE_DEVICE") == "1":
            local_rank = 0
        elif torch.cuda.device_count() < world:
            raise SystemExit(f"bench.py: --gpus {world} but only {torch.cuda.device_count()} device(s) are visible")
        torch.cuda.set_device(local_rank)
        be = os.environ.get("ZKHIP_BENCH_DIST_BACKEND", "nccl")
        if be == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(be)
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}")

    # the bulk communicator (csrc/comm.hip): off by default in the library since round 6 (a caller without a fallback ladder must not meet two
    # co-resident communicators first); bench.py opts in — its ladder's later rungs switch it off again (ZKHIP_COMM_BULK=0 in their environment)
    os.environ.setdefault("ZKHIP_COMM_BULK", "1")
    ctx = ffi.Context(local_rank)
    replay = args.replay_rank is not None
    shard = (world > 1 or replay) and not args.replicas and not (args.chain and args.agg_unsharded)
    nshare = args.of if replay else world        # ranks the sharded proof is split over
    replay_lib = None
    if args.row_sharded is not None:
        ctx.set_option("row_sharded", args.row_sharded)
    if args.msm_c is not None:
        ctx.set_option("msm_c", args.msm_c)
    if world > 1:
        ctx.set_option("comm_timeout_ms", args.comm_timeout_ms)
    if replay:
        import ctypes

        os.environ["ZKREPLAY_LATENCY_US"], os.environ["ZKREPLAY_LINK_GBS"] = str(args.replay_latency_us), str(args.replay_link_gbs)
        path = os.path.join(ROOT, "tools", "replay_rccl", "libreplay_rccl.so")
        if not os.path.exists(path):
            raise SystemExit(f"bench.py: {path} is missing (python -c 'import __graft_entry__ as g; g.build()')")
        ctx.comm_init_replay(args.replay_rank, args.of, path)
        replay_lib = ctypes.CDLL(path)      # the same loaded object (dlopen by path): its counters

        class ReplayStats(ctypes.Structure):
            _fields_ = [("collectives", ctypes.c_uint64), ("bytes_received", ctypes.c_uint64), ("bytes_sent", ctypes.c_uint64), ("wire_us", ctypes.c_double)]

        def replay_stats(reset=False):
            st = ReplayStats()
            replay_lib.ncclReplayStats(ctypes.byref(st), ctypes.c_int(1 if reset else 0))
            return dict(collectives=st.collectives, bytes_received=st.bytes_received, bytes_sent=st.bytes_sent, wire_us=st.wire_us)
        info = ctx.comm_describe()
        if info["nranks"] != args.of or info["transport_ranks"] != args.of or info["transport"] != "rccl":
            raise SystemExit(f"bench.py: the replay communicator reports {info}")
    elif shard:
        # RCCL communicator inside the library (unique id broadcast through torch.distributed).  If it cannot be created on some rank
        # (no librccl, an RCCL error) this worker FAILS — a line that says n_gpus N must come from N cooperating ranks — and the
        # supervisors move every rank to the next rung of the ladder in fresh processes (supervise()).
        hook = os.environ.get("ZKHIP_BENCH_FAIL_COMM", "")     # test hook: what a missing librccl / an RCCL error looks like ("1": always, "row": on the row-sharded rung only)
        if hook == "1" or (hook == "row" and os.environ.get("ZKHIP_ROW_SHARDED", "1") != "0"):
            raise SystemExit(f"bench.py rank {rank}: ZKHIP_BENCH_FAIL_COMM={hook} (injected communicator failure)")
        ctx.comm_init(rank, world, dist, transport=lib_transport())
        info = ctx.comm_describe()
        if info["nranks"] != world or info["transport_ranks"] not in (world, -1):
            raise SystemExit(f"bench.py: the library's communicator reports {info} for WORLD_SIZE {world}")
    bh = build_hash()
    gpu_proofs = []     # every distinct proof the HIP path made: what parity_check() compares with the CPU oracle's digests

    def note_proof(shape, kind, wseed, proof, where):
        gpu_proofs.append(dict(shape=shape.name, k=shape.k, transcript=kind, witness=wseed, where=where, proof_bytes=len(proof),
                               proof_sha256=hashlib.sha256(bytes(proof)).hexdigest()))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def teardown():
        """the line is out: leave.  Destroying communicators is collective-ish and has no deadline of its own, so a timer ends the process
        (exit code 0: the measurement is complete) if it has not returned in 30 s."""
        if world == 1:
            return
        import threading

        t_ = threading.Timer(30.0, lambda: os._exit(0))
        t_.daemon = True
        t_.start()
        dist.barrier()
        if shard:
            ctx.comm_destroy()
        dist.destroy_process_group()

    def run_config(name, steps, warmup, breakdown_passes=2, with_h2d=True, witness="uniform"):
        """-> result dict for one configuration (collective: every rank calls it with the same arguments)"""
        shape = make_shape(pv, name, args)
        kind = TRANSCRIPT[name]
        shard_mode = None
        if shard:
            shard_mode = args.shard if args.shard != "auto" else ("points" if shape.k >= 20 else "columns")
            ctx.comm_shard(shard_mode)
        torch.cuda.empty_cache()
        free0 = torch.cuda.mem_get_info()[0]
        t0 = time.perf_counter()
        backend = pv.GpuBackend(ctx, ffi)
        prover = pv.Prover(backend, shape, satisfiable=True)
        wseed = 0 if (shard or world == 1) else rank
        wit = prover.witness(wseed, dist=witness)
        torch.cuda.synchronize()
        setup_s = time.perf_counter() - t0
        n = 1 << shape.k
        counts = shape.counts(prover.dom.extended_k)
        en = 1 << prover.dom.extended_k
        # zkhip_create_proof_ex evaluates the quotient on quotient_poly_degree cosets of the size-n domain when that is fewer rows than the
        # extended domain (degree 4: 3 of 4); the Python schedule always uses the extended domain
        qd = prover.dom.quotient_poly_degree
        coset_q = qd if (not args.python_schedule and qd < (en >> shape.k) and os.environ.get("ZKHIP_COSET_QUOTIENT", "1") != "0") else 0
        q_rows = coset_q * n if coset_q else en
        prove = (lambda: prover.prove(wit, transcript=kind)) if args.python_schedule else (lambda: prover.prove_native(wit, transcript=kind))
        t1 = time.perf_counter()
        prove()                      # the first proof of a fresh process: what one run of a reference command pays (setup_s + this)
        torch.cuda.synchronize()
        first_s = time.perf_counter() - t1
        for _ in range(warmup - 1):
            prove()
        resident = free0 - torch.cuda.mem_get_info()[0]
        # Live HIP-event timing inside the timed region covers the dominant kernel only (every recorded span costs two event records
        # on the launch stream); the full per-kernel breakdown comes from extra, untimed passes afterwards.
        DOMINANT = "msm_accum_affine"
        ctx.profile_select(DOMINANT)
        ctx.profile_enable(True)
        g0 = ctx.comm_bytes_gathered()
        c0 = ctx.comm_describe()["collectives"] if shard else 0
        barrier()
        if replay:
            replay_stats(reset=True)
        t0 = time.perf_counter()
        native_s = 0.0
        for _ in range(steps):
            trace = prove()
            native_s += trace.get("native_call_s", 0.0)
        barrier()
        dt = time.perf_counter() - t0
        rstats = {k_: v_ / steps for k_, v_ in replay_stats().items()} if replay else None
        gathered = (ctx.comm_bytes_gathered() - g0) // max(steps, 1)
        coll_per_proof = (ctx.comm_describe()["collectives"] - c0) / max(steps, 1) if shard else 0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        ms_per_step = dt * 1000.0 / steps
        ms, launches = ctx.profile_read(DOMINANT)
        live = dict(ms_per_step=ms / steps, launches_per_step=launches / steps)
        kernels = {}
        ctx.profile_select(None)
        ctx.profile_enable(True)
        for _ in range(breakdown_passes):
            prove()
        for kname in KERNELS:
            ms, launches = ctx.profile_read(kname)
            kernels[kname] = dict(ms_per_step=round(ms / breakdown_passes, 4), launches_per_step=launches / breakdown_passes)
        real_pairs = ctx.profile_counter("msm_pairs") / breakdown_passes          # non-zero digits only (this rank's share)
        dense_pairs = ctx.profile_counter("msm_dense_pairs") / breakdown_passes    # n x windows per column
        trace_passes = None
        if replay:
            # the rank's exchange timeline (zkhip_comm_trace): when each of the proof's exchanges completed on the communicator's stream, relative to the
            # proof's start, in 3 extra untimed passes — what tools/install_rank_replay.py composes into a synchronised step (sum over exchanges of the
            # max over ranks of the span between consecutive exchanges) beside the max over ranks of whole proofs
            ctx.profile_enable(False)
            trace_passes = []
            for _ in range(3):
                ctx.comm_trace(True)
                prove()
                ent, end_us = ctx.comm_trace_read()
                trace_passes.append(dict(end_us=round(end_us, 1), done_us=[e["stream_done_us"] for e in ent], host_issue_us=[e["host_issue_us"] for e in ent],
                                         exchanges=[[e["phase"], e["kind"], int(e["bulk"]), e["bytes_received"]] for e in ent]))
            ctx.comm_trace(False)
        # the NTT and sweep kernels overlap the MSM phases inside a proof (two streams), so their in-proof event spans are stretched
        # by the kernels they share the chip with: their rooflines are taken from isolated launches of the same shapes instead
        iso = {}
        if not shard:
            dom = prover.dom
            batch = [ctx.synth_fill(n, 9000 + j) for j in range(8)]
            # the transform the proof issues: onto q cosets of the size-n domain when that is fewer rows (csrc/cosets.hip), else onto the extended domain
            to_quotient_domain = dom.coeff_to_cosets_device if coset_q else dom.coeff_to_extended_device
            to_quotient_domain(batch)
            ctx.profile_enable(True)
            reps = 3
            for _ in range(reps):
                outs = to_quotient_domain(batch)
            ms_s, l_s = ctx.profile_read("ntt_strided")
            ms_f, l_f = ctx.profile_read("ntt_final")
            iso["ntt"] = dict(ms=(ms_s + ms_f) / reps, launches=(l_s + l_f) / reps, elems=8 * q_rows, per_kernel=dict(ntt_strided=(ms_s / reps, l_s / reps), ntt_final=(ms_f / reps, l_f / reps)))
            ctx.profile_enable(False)
            del batch, outs
        ctx.profile_enable(False)
        h2d = None
        if with_h2d and world == 1 and not replay and not args.python_schedule and not args.no_h2d:
            # the same step with the advice columns handed over as pinned HOST arrays (a Rust caller's Vec<Fr> columns) and the
            # instance columns built from the instance values: the uploads are inside the timed region
            prover.prove_native(wit, transcript=kind, host_inputs=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            hs = max(3, steps // 2)
            for _ in range(hs):
                prover.prove_native(wit, transcript=kind, host_inputs=True)
            torch.cuda.synchronize()
            def timed_host(mode):
                prover.prove_native(wit, transcript=kind, host_inputs=mode)
                torch.cuda.synchronize()
                t_ = time.perf_counter()
                for _ in range(hs):
                    prover.prove_native(wit, transcript=kind, host_inputs=mode)
                torch.cuda.synchronize()
                return round((time.perf_counter() - t_) / hs, 6)
            pinned_s = round((time.perf_counter() - t0) / hs, 6)
            # the same from PAGEABLE host arrays — what a Rust caller's Vec<Fr> columns are: the library issues those copies from a worker thread (option host_copy_thread,
            # round 6), so that they block that thread and not the one launching the proof; with the worker switched off every copy blocks the proof's thread for its own duration
            pageable_s = timed_host("pageable")
            ctx.set_option("host_copy_thread", 0)
            unthreaded_s = timed_host("pageable")
            ctx.set_option("host_copy_thread", 1)
            wit.pop("advice_host_pageable", None)
            h2d = dict(value=pinned_s, pageable_value=pageable_s, pageable_without_copy_thread_value=unthreaded_s, unit="s", steps=hs, h2d_bytes=shape.n_advice * n * 32,
                       note="advice columns uploaded from pinned host memory inside the step (copy stream; the random polynomial's commitment and, for many-column circuits, the earlier column groups' commitments overlap the uploads); never part of `value`")
        barrier()

        # ---- rooflines (algorithmic bytes: SURVEY.md §8(d)); one rank's share when the proof is sharded
        share = nshare if shard else 1
        c_bits, windows = backend.params.window()
        traffic, traffic_file = pmc_traffic(name, bh) if (world == 1 and not replay) else (None, None)
        def per_launch(kname):
            k_ = kernels[kname]
            return (k_["ms_per_step"] / k_["launches_per_step"]) if k_["launches_per_step"] else 0.0
        roof = {}
        # MSM accumulation (live figure): 96 B per (scalar, point) pair
        pairs = counts["msm"] * n // share
        a_l = max(live["launches_per_step"], 1)
        alg = 96.0 * pairs / a_l
        avg_s = live["ms_per_step"] / a_l / 1000.0
        ach = alg / avg_s / 1e9 if avg_s > 0 else 0.0
        # one XYZZ mixed addition (8 products + 2 squarings, 9 reductions) per (non-zero digit, point) pair: zero digits are skipped, so the count is
        # the one the library reports from its sort (dense = n x windows per column; bit / small-valued columns have far fewer)
        # v_mad_u64_u32 per addition: 8 products x 81 + 2 squarings x 45 + 9 Montgomery reductions x 90 (r (Q - X3) - Y1 PPP shares one)
        mads = real_pairs * (8 * 81 + 2 * 45 + 9 * 90)
        int_ach = mads / (live["ms_per_step"] / 1000.0) / 1e12 if live["ms_per_step"] > 0 else 0.0
        roof["msm_accum_affine"] = {"kernel": "k_accum_affine", "bound": "valu", "hbm_frac": round(ach / HBM_PEAK_GBS, 5), "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": round(traffic["msm_accum_affine"]) if traffic and traffic["msm_accum_affine"] else None,
                                    "algorithmic_bytes_per_launch": round(alg), "avg_launch_ms": round(avg_s * 1000.0, 4), "timing": "HIP events inside the timed region",
                                    "int_roofline": {"bound": "v_mad_u64_u32 issue", "achieved": round(int_ach, 2), "peak": MAD_PEAK_T, "unit": "Tmad/s",
                                                     "frac": round(int_ach / MAD_PEAK_T, 4), "window_bits": c_bits, "windows": windows,
                                                     "digit_pairs_per_step": round(real_pairs), "dense_digit_pairs_per_step": round(dense_pairs),
                                                     "peak_source": "SELF-MEASURED, not a vendor figure: tools/microbench (dense v_mad_u64_u32 loop, ILP 8, 8 waves per SIMD) = "
                                                                    "29.87 Tmad/s in profiles/r02_microbench_gfx950.txt:20 (5.27 cycles per wave-instruction per SIMD; "
                                                                    "4.6 by profiles/r04_clock_probe_gfx950.txt at the sustained 2.37 GHz)"},
                                    "note": "VALU-issue (integer multiply) bound: valu_busy 0.89-0.99 in profiles/r04_v7_valu_*.csv; the window tables trade HBM bytes for doublings (counted traffic 6-18x the algorithmic bytes): see DESIGN.md 5"}
        # NTT: 64 B per element per transform (one read + one write), whatever the number of passes.  Isolated batch of 8 coset NTTs
        # (coeff_to_extended: n -> extended_n) — the shape the proof issues
        ntt_ms = kernels["ntt_strided"]["ms_per_step"] + kernels["ntt_final"]["ms_per_step"]
        if "ntt" in iso and iso["ntt"]["ms"] > 0:
            i_ = iso["ntt"]
            ach = 64.0 * i_["elems"] / (i_["ms"] / 1000.0) / 1e9
            tr_l = None
            if traffic and traffic["ntt_strided"] and traffic["ntt_final"]:
                tr_l = (traffic["ntt_strided"] * i_["per_kernel"]["ntt_strided"][1] + traffic["ntt_final"] * i_["per_kernel"]["ntt_final"][1]) / max(i_["launches"], 1)
            roof["ntt"] = {"kernel": (traffic or {}).get("ntt_kernels") or ("k_ntt_strided_r4 + k_ntt_final_r4" if shape.k >= 21 else "k_ntt_strided_r4s + k_ntt_final_r4s"), "bound": "valu", "hbm_frac": round(ach / HBM_PEAK_GBS, 5), "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": round(tr_l) if tr_l else None,
                           "algorithmic_bytes_per_launch": round(64.0 * i_["elems"] / max(i_["launches"], 1)), "avg_launch_ms": round(i_["ms"] / max(i_["launches"], 1), 4),
                           "timing": ("HIP events, 3 isolated batches of 8 columns onto %d cosets (8 x %d transforms of 2^%d) after a warm-up" % (coset_q, coset_q, shape.k)) if coset_q
                                     else "HIP events, 3 isolated batches of 8 coset NTTs (2^%d -> 2^%d) after a warm-up" % (shape.k, prover.dom.extended_k),
                           "in_proof_ms_per_step": round(ntt_ms, 3),
                           "transforms_per_step": counts["intt_n"] + counts["ntt_ext"] * (coset_q or 1) + counts["intt_ext"] * (coset_q or 1),
                           "note": "algorithmic = 64 B per element per transform; a transform of 2^m elements is ceil(m / 9) launches; VALU-issue bound "
                                   "(4 elements per thread, four waves per SIMD; valu_busy in profiles/r04_v7_valu_*.csv; the memory side alone is 0.6 of the time and the two overlap imperfectly: profiles/r03_ntt_experiments.md), not HBM bound; in-proof spans overlap the MSM phases"}
        # sweep: 32 B x (distinct (column, rotation) reads + 1 write) per extended row
        sw = kernels["sweep"]
        if sw["ms_per_step"] > 0 and not shard:
            reads = len(shape.queries()) + len(shape.perm_columns) + 2 * shape.n_perm_sets + 5 * len(shape.lookups) + 3
            alg = 32.0 * (reads + 1) * q_rows
            main_ms = per_launch("sweep")
            # the lookup compressions run through the same kernel on 2^k rows; the quotient sweep is the one long launch
            ach = alg / (sw["ms_per_step"] / 1000.0) / 1e9
            roof["sweep"] = {"kernel": "k_sweep", "bound": "valu", "hbm_frac": round(ach / HBM_PEAK_GBS, 5), "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
                             "traffic": round(traffic["sweep"]) if traffic and traffic["sweep"] else None, "algorithmic_bytes_per_launch": round(alg / max(sw["launches_per_step"], 1)),
                             "avg_launch_ms": round(main_ms, 4), "distinct_reads_per_row": reads, "timing": f"HIP events, {breakdown_passes} untimed passes"}
        res = {"value": round(ms_per_step / 1000.0, 6), "unit": "s", "steps": steps, "warmup": warmup, "ms_per_step": round(ms_per_step, 3),
               "workload": f"{shape.name}: {counts['msm']} MSM_2^{shape.k} + {counts['intt_n']} iNTT_2^{shape.k} + " + (
                               f"{counts['ntt_ext']} x {coset_q} coset NTT_2^{shape.k} + {coset_q} iNTT_2^{shape.k} + sweep over {coset_q} x 2^{shape.k} rows (the quotient on {coset_q} of the "
                               f"{1 << (prover.dom.extended_k - shape.k)} cosets of the 2^{prover.dom.extended_k} extended domain)" if coset_q else
                               f"{counts['ntt_ext']} NTT_2^{prover.dom.extended_k} + 1 iNTT_2^{prover.dom.extended_k} + sweep over 2^{prover.dom.extended_k} rows") +
                           f" + lookup permute, {shape.n_perm_sets}+{len(shape.lookups)} grand products, "
                           f"evaluations, SHPLONK; satisfiable synthetic instance (valid proof), free witness cells "
                           + ("uniform field elements (the worst case for the commitments)" if witness == "uniform" or shape.layout == "sha" else
                              "drawn from SURVEY.md 8(d)'s value mix (limbs / bits / uniform)")
                           + (" [bit / word columns by construction]" if shape.layout == "sha" else "") + f"; {kind} transcript as {REFERENCE_CMD[name]}",
               "workload_short": f"{shape.name}: {counts['msm']} MSM + {counts['intt_n']} iNTT + " + (f"{counts['ntt_ext']}x{coset_q} coset NTT + {coset_q} iNTT" if coset_q else
                                 f"{counts['ntt_ext']} NTT_2^{prover.dom.extended_k} + 1 iNTT_2^{prover.dom.extended_k}") + f" of 2^{shape.k}, sweep {q_rows >> shape.k}x2^{shape.k} rows, SHPLONK; {kind}",
               "witness": "bits / words (SHA-256 bit circuit layout)" if shape.layout == "sha" else witness,
               "k": shape.k, "advice": shape.n_advice, "fixed": shape.n_fixed, "instance_values": prover.n_instance_values, "lookups": len(shape.lookups),
               "perm_columns": len(shape.perm_columns), "degree": shape.degree, "transcript": kind, "proof_bytes": len(trace.get("proof", b"")),
               "proof_sha256": hashlib.sha256(bytes(trace.get("proof", b""))).hexdigest(),
               "setup_s": round(setup_s, 3), "resident_bytes": int(resident), "rooflines": roof, "kernels_ms_per_step": kernels,
               "traffic_source": traffic_file, "with_h2d": h2d, "msm_shard": shard_mode,
               "first_proof_s": round(setup_s + first_s, 3), "comm": comm_fields(gathered, shard_mode, coll_per_proof),
               "native_call_ms_per_step": round(native_s * 1000.0 / steps, 3) if native_s else None}   # zkhip_create_proof_ex alone; ms_per_step also holds the ctypes wrapper around it
        if replay:
            res["replay_exchanges_per_step"] = rstats
            res["replay_trace"] = trace_passes
        if not replay:      # a replayed rank's bytes are wrong by construction: never offered for comparison
            note_proof(shape, kind, wseed if witness == "uniform" else f"{witness}:{wseed}", trace.get("proof", b""), f"configs.{name}" + ("" if witness == "uniform" else f" ({witness} witness)"))
        prover.release()          # the context's per-key caches (coset-layout key columns, sorted lookup table)
        backend.params.free()
        del prover, wit, trace, backend
        # the prover object sits in reference cycles: collect it NOW — left to the cyclic collector, the hipFree / hipHostFree of its 5.5 GiB of
        # columns and 512 MiB of pinned upload buffers ran somewhere inside the NEXT configuration's timed set-up (0.6 s of rsa17's setup_s, r03)
        import gc

        gc.collect()
        torch.cuda.empty_cache()
        return res, shape

    def parity_sample(name, k_):
        """ONE proof of the headline shape at the size the CPU leg's bounded sample is timed at (default k = 20 for the k = 22 headline): the
        default line then carries a byte comparison with the CPU oracle on the full circuit shape, through the same (sharded, when N > 1)
        path, without the two-minute CPU pass at k = 22.  Collective: every rank calls it."""
        shape = make_shape(pv, name, argparse.Namespace(**{**vars(args), "agg_k": k_}))
        if shard:
            ctx.comm_shard(args.shard if args.shard != "auto" else ("points" if shape.k >= 20 else "columns"))
        backend = pv.GpuBackend(ctx, ffi)
        prover = pv.Prover(backend, shape, satisfiable=True)
        wit = prover.witness(0)
        t0 = time.perf_counter()
        pf = bytes(prover.prove_native(wit, transcript=TRANSCRIPT[name])["proof"])
        torch.cuda.synchronize()
        first_ms = (time.perf_counter() - t0) * 1000.0
        if bytes(prover.prove_native(wit, transcript=TRANSCRIPT[name])["proof"]) != pf:
            raise SystemExit(f"bench.py: the k = {k_} parity sample's proof is not deterministic")
        note_proof(shape, TRANSCRIPT[name], 0, pf, f"parity sample (the headline shape at the CPU sample's size, k = {k_})")
        prover.release()
        backend.params.free()
        del prover, wit, backend
        import gc

        gc.collect()
        torch.cuda.empty_cache()
        return dict(k=k_, shape=shape.name, transcript=TRANSCRIPT[name], proof_bytes=len(pf), proof_sha256=gpu_proofs[-1]["proof_sha256"], first_proof_ms=round(first_ms, 2))

    def ffi_levels(name, one_call_s, reps=3):
        """The drop-in boundary at each patch level: how long ONE PROOF'S WORTH of the calls a patch level
        replaces takes through the entry points that level binds, with the caller's data in pageable host memory (a Rust Vec<Fr>) — one call per
        commitment / transform, in sequence, exactly as upstream's create_proof issues them (/root/reference/src/helpers.rs:233,299, src/bin/cli.rs:320,369,519).
          curves   halo2curves patched only: counts.msm x zkhip_msm_g1 + (counts.intt_n x best_fft(2^k) + (counts.ntt_ext + counts.intt_ext) x best_fft(2^extended_k))
                   through zkhip_fft.  Everything else of create_proof (sweep, permute, grand products, evaluations ...) stays upstream's CPU code: NOT in the figure.
          domain   + EvaluationDomain patched: the transforms through zkhip_lagrange_to_coeff / zkhip_coeff_to_extended / zkhip_extended_to_coeff (host forms).
          one-call zkhip_create_proof_ex: the whole proof (the headline)."""
        import numpy as np

        which = ("curves", "domain", "one-call") if args.ffi_level == "all" else (args.ffi_level,)
        shape = make_shape(pv, name, args)
        backend = pv.GpuBackend(ctx, ffi)
        prover = pv.Prover(backend, shape, satisfiable=True)
        dom, params = prover.dom, backend.params
        n, en = 1 << shape.k, 1 << dom.extended_k
        counts = shape.counts(dom.extended_k)
        cols = [ctx.to_host(ctx.synth_fill(n, 7000 + j)).copy() for j in range(4)]      # numpy-owned: pageable
        ext = np.empty((en, 4), dtype=np.uint64)
        ext[:] = np.tile(cols[0], (en // n, 1))
        out = {}

        def timed(f):
            f()
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                f()
                ts.append(time.perf_counter() - t0)
            return statistics.median(ts)

        def call(fn, *a):
            ffi._check(getattr(ffi.lib(), fn)(ctx.h, *a))
        import ctypes as C_

        t_msm = timed(lambda: [params.commit(cols[j % 4]) if j % 2 else params.commit_lagrange(cols[j % 4]) for j in range(counts["msm"])])
        per = {"zkhip_msm_g1_ms": round(t_msm * 1000.0 / counts["msm"], 3)}
        if "curves" in which:
            w_n, w_e = ffi._u64(dom.omega), ffi._u64(dom.extended_omega)
            t_fn = timed(lambda: [call("zkhip_fft", ffi._p(cols[j % 4]), ffi._p(w_n), C_.c_uint32(shape.k)) for j in range(counts["intt_n"])])
            ne = counts["ntt_ext"] + counts["intt_ext"]
            t_fe = timed(lambda: [call("zkhip_fft", ffi._p(ext), ffi._p(w_e), C_.c_uint32(dom.extended_k)) for _ in range(ne)])
            per.update(zkhip_fft_n_ms=round(t_fn * 1000.0 / counts["intt_n"], 3), zkhip_fft_extended_ms=round(t_fe * 1000.0 / ne, 3))
            out["curves"] = dict(value=round(t_msm + t_fn + t_fe, 6), unit="s", calls={"zkhip_msm_g1 (2^%d)" % shape.k: counts["msm"], "zkhip_fft (2^%d)" % shape.k: counts["intt_n"],
                                 "zkhip_fft (2^%d)" % dom.extended_k: ne}, parts_s=dict(msm=round(t_msm, 6), fft_n=round(t_fn, 6), fft_extended=round(t_fe, 6)),
                                 note="GPU-side calls only: the rest of create_proof is upstream's CPU code at this patch level and is not in the figure")
        if "domain" in which:
            t_l = timed(lambda: [call("zkhip_lagrange_to_coeff", dom.h, ffi._p(cols[j % 4])) for j in range(counts["intt_n"])])
            t_c = timed(lambda: [call("zkhip_coeff_to_extended", dom.h, ffi._p(cols[j % 4]), C_.c_size_t(n), ffi._p(ext)) for j in range(counts["ntt_ext"])])
            t_x = timed(lambda: [call("zkhip_extended_to_coeff", dom.h, ffi._p(ext)) for _ in range(counts["intt_ext"])])
            per.update(zkhip_lagrange_to_coeff_ms=round(t_l * 1000.0 / counts["intt_n"], 3), zkhip_coeff_to_extended_ms=round(t_c * 1000.0 / counts["ntt_ext"], 3),
                       zkhip_extended_to_coeff_ms=round(t_x * 1000.0 / counts["intt_ext"], 3))
            out["domain"] = dict(value=round(t_msm + t_l + t_c + t_x, 6), unit="s", calls={"zkhip_msm_g1 (2^%d)" % shape.k: counts["msm"], "zkhip_lagrange_to_coeff": counts["intt_n"],
                                 "zkhip_coeff_to_extended (2^%d -> 2^%d)" % (shape.k, dom.extended_k): counts["ntt_ext"], "zkhip_extended_to_coeff": counts["intt_ext"]},
                                 parts_s=dict(msm=round(t_msm, 6), lagrange_to_coeff=round(t_l, 6), coeff_to_extended=round(t_c, 6), extended_to_coeff=round(t_x, 6)),
                                 note="GPU-side calls only: sweep, permute, grand products, evaluations, SHPLONK stay upstream's CPU code at this patch level")
        if "one-call" in which:
            out["one-call"] = dict(value=one_call_s, unit="s", calls={"zkhip_create_proof_ex": 1}, note="the whole proof, transcript included (the headline)")
        # the device-resident one-column forms of the same two kernels, for the ratio the host-pointer level is judged by
        d_col = ctx.synth_fill(n, 7100)

        def dev_msm():
            params.commit_batch_device([d_col])
            ctx.synchronize()
        per["msm_device_resident_one_column_ms"] = round(timed(dev_msm) * 1000.0, 3)
        per["zkhip_msm_g1_over_device_resident"] = round(per["zkhip_msm_g1_ms"] / per["msm_device_resident_one_column_ms"], 3)
        out["per_call"] = per
        out["host_memory"] = "pageable (numpy-owned arrays), one blocking call per commitment / transform"
        prover.release()
        backend.params.free()
        del prover, backend, cols, ext, d_col
        import gc

        gc.collect()
        torch.cuda.empty_cache()
        return out

    def run_chain(steps, warmup):
        """BASELINE configs[4] (/root/reference/src/tests/x509_aggregation.rs:20-110): four independent leaf proofs (rsa, sha, rsa, sha), a
        barrier, then the aggregation proof.  N >= 4: one leaf proof per rank 0..3 on an unsharded context, then the k = 22 proof
        sharded over all N ranks; N = 1: the five proofs one after the other.  (The aggregation circuit's witness generation — the
        in-circuit verification of the four snarks on the CPU, src/lib.rs:43-49 — is outside the path and not timed.)  Collective."""
        vworld, vrank = (nshare, args.replay_rank) if replay else (world, rank)      # (--replay-rank: the rank this process plays, alone)
        if vworld > 1 and vworld < 4:
            raise SystemExit("--chain needs --gpus 1 or >= 4")
        agg_here = shard or vworld == 1 or vrank == 0        # --agg-unsharded (last rung of the ladder): the aggregation proof on rank 0 alone
        leaf_ctx = ffi.Context(local_rank) if shard else ctx
        leaf_names = ["rsa17", "sha19", "rsa17", "sha19"]
        # Who proves which leaf.  Leaf j's head is rank j; from N = 6 on the ranks 4.. — idle until the aggregation proof when every leaf sits on
        # one rank, while the SHA-shaped leaves take 4.5x as long as the RSA ones — are dealt to the two SHA leaves in turn (N = 8: leaf 1 over
        # ranks 1 + 4, leaf 3 over ranks 3 + 5): such a leaf is ONE proof over its group's own communicator, MSMs by column (k = 19: one MSM
        # cannot fill several GPUs), the same bytes on every member.  The leaf contexts' communicators are used before the barrier, the
        # aggregation proof's after it: never two collectives of different communicators in flight on one device.
        groups = chain_leaf_groups(vworld, shard and args.leaf_groups and not args.no_leaf_groups)
        my_leaf = next((j for j, rs in groups.items() if vrank in rs), None)
        if any(len(rs) > 1 for rs in groups.values()):
            pgs = {}
            if not replay:
                for j in sorted(groups):      # (new_group is collective over ALL ranks, in the same order everywhere)
                    if len(groups[j]) > 1:
                        pgs[j] = dist.new_group(groups[j])
            if my_leaf is not None and len(groups[my_leaf]) > 1:
                rs = groups[my_leaf]
                if world > 1:
                    leaf_ctx.set_option("comm_timeout_ms", args.comm_timeout_ms)
                if replay:
                    leaf_ctx.comm_init_replay(rs.index(vrank), len(rs), os.path.join(ROOT, "tools", "replay_rccl", "libreplay_rccl.so"))
                else:
                    leaf_ctx.comm_init(rs.index(vrank), len(rs), dist, transport=lib_transport(), group=pgs[my_leaf], src=rs[0])
                leaf_ctx.comm_shard("columns")
        mine = list(enumerate(leaf_names)) if vworld == 1 else ([(my_leaf, leaf_names[my_leaf])] if my_leaf is not None else [])
        leaves, pairs = [], 0.0
        for j, nm in mine:
            sh_ = make_shape(pv, nm, args)
            pr_ = pv.Prover(pv.GpuBackend(leaf_ctx, ffi), sh_, satisfiable=True)
            leaves.append((pr_, dict(pr_.witness(j), seed=j), TRANSCRIPT[nm]))      # witness j: the same instance on every member of leaf j's group
            pairs += sh_.counts(pr_.dom.extended_k)["msm"] * float(1 << sh_.k) / len(groups[j] if vworld > 1 else [0])
        if shard:
            ctx.comm_shard("points" if args.shard == "auto" else args.shard)
        agg = agg_w = None
        if agg_here:
            agg_shape = make_shape(pv, "agg22", args)
            agg = pv.Prover(pv.GpuBackend(ctx, ffi), agg_shape, satisfiable=True)
            agg_w = agg.witness(0)
            pairs += agg_shape.counts(agg.dom.extended_k)["msm"] * float(1 << agg_shape.k) / (nshare if shard else 1)
        sizes, digests, last = [], [], []
        phase_s = dict(leaf=0.0, agg=0.0)

        def chain_step(trace=None):
            """trace: a list that receives the aggregation proof's exchange timeline (--replay-rank: zkhip_comm_trace)"""
            del sizes[:], digests[:], last[:]
            t_l = time.perf_counter()
            for pr_, w_, kind_ in leaves:
                pf_ = bytes(pr_.prove_native(w_, transcript=kind_)["proof"])
                sizes.append(len(pf_))
                digests.append(hashlib.sha256(pf_).hexdigest())
                last.append((pr_.shape, kind_, w_["seed"], pf_, f"chain leaf (witness {w_['seed']})"))
            barrier()
            t_a = time.perf_counter()
            phase_s["leaf"] += t_a - t_l
            if agg_here:
                if trace is not None:
                    ctx.comm_trace(True)
                pf_ = bytes(agg.prove_native(agg_w, transcript="evm")["proof"])
                if trace is not None:
                    ent, end_us = ctx.comm_trace_read()
                    trace.append(dict(end_us=round(end_us, 1), done_us=[e["stream_done_us"] for e in ent], host_issue_us=[e["host_issue_us"] for e in ent],
                                      exchanges=[[e["phase"], e["kind"], int(e["bulk"]), e["bytes_received"]] for e in ent]))
                sizes.append(len(pf_))
                digests.append(hashlib.sha256(pf_).hexdigest())
                last.append((agg.shape, "evm", 0, pf_, "chain aggregation proof"))
            phase_s["agg"] += time.perf_counter() - t_a

        for _ in range(warmup):
            chain_step()
        DOMINANT = "msm_accum_affine"
        ctxs = [ctx] + ([leaf_ctx] if leaf_ctx is not ctx else [])
        for c_ in ctxs:
            c_.profile_select(DOMINANT)
            c_.profile_enable(True)
        g0 = ctx.comm_bytes_gathered()
        c0 = ctx.comm_describe()["collectives"] if shard else 0
        barrier()
        phase_s["leaf"] = phase_s["agg"] = 0.0
        if replay:
            replay_stats(reset=True)
        t0 = time.perf_counter()
        for _ in range(steps):
            chain_step()
        barrier()
        dt = time.perf_counter() - t0
        rstats = {k_: v_ / steps for k_, v_ in replay_stats().items()} if replay else None
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        timed_phase_s = dict(phase_s)      # (the traced steps below keep adding to phase_s: the line reports the TIMED steps' phases)
        agg_trace = None
        if replay and shard and agg_here:      # the aggregation proof's exchange timeline on this rank, 3 extra untimed steps
            agg_trace = []
            for _ in range(3):
                chain_step(trace=agg_trace)
            ctx.comm_trace(False)
        acc_ms, acc_l = 0.0, 0
        for c_ in ctxs:
            ms_, l_ = c_.profile_read(DOMINANT)
            acc_ms, acc_l = acc_ms + ms_, acc_l + l_
            c_.profile_select(None)
            c_.profile_enable(False)
        roof = None
        if acc_l and acc_ms > 0:
            alg = 96.0 * pairs * steps / acc_l
            avg_s = acc_ms / acc_l / 1000.0
            ach = alg / avg_s / 1e9
            roof = {"kernel": "k_accum_affine", "bound": "valu", "hbm_frac": round(ach / HBM_PEAK_GBS, 5), "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": None, "algorithmic_bytes_per_launch": round(alg), "avg_launch_ms": round(avg_s * 1000.0, 4),
                    "launches_per_step": acc_l / steps, "timing": "HIP events inside the timed region, all proofs of the chain this rank ran",
                    "note": "rank 0's share: 96 B per (scalar, point) pair over every commitment of its leaf proof(s) and its 1/N of the aggregation proof's; "
                            "VALU-issue bound like the single proofs (DESIGN.md 5); traffic: see the per-configuration PMC passes of the N = 1 line"}
        # every rank's leaf digest to rank 0: the members of a leaf's group must hold the same bytes, and the line carries all four leaves'
        leaf_digests = {str(w_["seed"]): d_ for (pr_, w_, _), d_ in zip(leaves, digests)}
        if world > 1:
            got = [None] * world
            dist.all_gather_object(got, leaf_digests)
            leaf_digests = {}
            for r_, g_ in enumerate(got):
                for j_, d_ in (g_ or {}).items():
                    if leaf_digests.setdefault(j_, d_) != d_:
                        raise SystemExit(f"bench.py: rank {r_} holds another proof of leaf {j_} than a lower rank of its group")
        if vworld == 1:
            par = "5 proofs in sequence on 1 GPU"
        elif shard:
            par = ("leaf proofs " + ", ".join(f"{leaf_names[j]} on rank{'s' if len(rs) > 1 else ''} {'+'.join(map(str, rs))}" for j, rs in sorted(groups.items()))
                   + (" (a leaf over several ranks: one proof on the group's own communicator, MSMs by column)" if any(len(rs) > 1 for rs in groups.values()) else "")
                   + f", barrier, then one proof sharded x{vworld}")
        else:
            par = "leaf proofs on ranks 0-3 (one each), then the aggregation proof on rank 0 alone (no collective on the data path)"
        res = {"value": round(dt / steps, 6), "unit": "s", "steps": steps, "warmup": warmup, "ms_per_step": round(dt * 1000.0 / steps, 3), "proofs_per_step": 5,
               "workload": "chain (BASELINE configs[4]): 2 x RSA k=17 + 2 x SHA256-shaped k=19 leaf proofs (Poseidon), barrier, aggregation-shaped "
                           f"k={args.agg_k} proof (Keccak)",
               "parallelism": par, "proof_bytes": list(sizes), "proof_sha256": list(digests), "roofline": roof,
               "phase_ms_per_step": {"leaf_proofs_until_the_barrier": round(timed_phase_s["leaf"] * 1000.0 / steps, 3), "aggregation_proof": round(timed_phase_s["agg"] * 1000.0 / steps, 3),
                                     "note": "host wall clock on this rank; the barrier (and a device synchronize) separates the two phases"},
               "replay_exchanges_per_step": rstats, "replay_trace": agg_trace, "leaf_proof_sha256": leaf_digests, "leaf_groups": {str(j): rs for j, rs in sorted(groups.items())},
               "bytes_gathered_per_step": (ctx.comm_bytes_gathered() - g0) // steps if shard else 0,
               "collectives_per_step": (ctx.comm_describe()["collectives"] - c0) / steps if shard else 0}
        for item in ([] if replay else last):
            note_proof(*item)
        for pr_, _, _ in leaves:
            pr_.release()
            pr_.b.params.free()
        if agg is not None:
            agg.release()
            agg.b.params.free()
        if leaf_ctx is not ctx:
            if leaf_ctx.world > 1 and not replay:
                leaf_ctx.comm_destroy()
            leaf_ctx.close()
        return res

    def comm_fields(per_step_bytes, shard_mode, collectives_per_step=None):
        if not shard:
            return None
        d = ctx.comm_describe()
        return {"transport": d["transport"], "nranks": d["nranks"], "transport_ranks": d["transport_ranks"], "collectives_total": d["collectives"],
                "collectives_per_step": collectives_per_step,     # exchanges (all-gathers + all-to-alls) the library issued per timed step on this rank
                "bytes_gathered_per_step": int(per_step_bytes), "shard_mode": shard_mode,
                "exchange_modes": {k_: ctx.profile_counter(k_) for k_ in ("proofs_row_sharded", "proofs_pieces_sharded", "shplonk_row_sharded")},
                "bulk_communicator": bool(ctx.profile_counter("comm_bulk")), "collectives_bulk_total": ctx.profile_counter("collectives_bulk"),   # the row windows' own communicator (comm.hip)
                "note": "nranks / transport_ranks / bytes: what the library's own communicator reports on rank 0 (zkhip_comm_info / zkhip_comm_describe; "
                        "transport_ranks = ncclCommCount); bytes = received by this rank through all-gathers in one step"}

    def replay_block(exchanges, trace=None):
        """what the line says about a --replay-rank run"""
        d = ctx.comm_describe()
        return {"rank": args.replay_rank, "of": args.of, "exchanges_per_step": exchanges, "trace": trace, "library_counters": {"collectives_total": d["collectives"], "bytes_gathered_total": d["bytes_gathered"]},
                "modelled_wire": {"latency_us_per_exchange": args.replay_latency_us, "link_GBps_per_peer_and_direction": args.replay_link_gbs,
                                  "note": "0 / 0: an exchange costs only the kernel that fabricates what would arrive (HBM write speed); otherwise the communicator's stream is "
                                          "additionally held for latency + max-over-peers(bytes from that peer) / link bandwidth per exchange (xGMI: one link per peer)"},
                "note": f"SINGLE-RANK REPLAY, not an {args.of}-GPU measurement: this process ran exactly what rank {args.replay_rank} of {args.of} runs (csrc/comm.hip's RCCL branch "
                        "against tools/replay_rccl: the peers' partial sums, rows and evaluations are fabricated on the device), alone on one GPU.  `value` is that rank's "
                        "time per step; the proof bytes are wrong by construction and are not compared with anything.  exchanges_per_step: what the stand-in "
                        "saw on the RCCL branch (silent pairs skipped, no padding): collectives, bytes this rank would receive / send, modelled wire time"}

    if args.chain:
        res = run_chain(args.steps, args.warmup)
        # one GPU: the five proofs in sequence, nothing scales (null); N >= 4: total work fixed while the aggregation proof is sharded ("strong");
        # --agg-unsharded (last rung of the ladder): the aggregation proof on rank 0 alone — more GPUs do not shorten it ("none")
        chain_scaling = (None if replay else "strong") if world == 1 else ("strong" if shard else "none")      # (a --replay-rank run is one process: null)
        if rank == 0:
            cb = parity = None
            if world == 1 and not args.no_cpu_baseline:      # N > 1: rank 0's GPU-free supervisor times it once the workers are gone
                try:
                    cb = chain_cpu_baseline(pv, args)
                    parity = parity_check(gpu_proofs)
                except Exception as e:   # noqa: BLE001 — the GPU measurement stands on its own
                    cb = dict(error=f"{type(e).__name__}: {e}"[:300])
            emit({"metric": "create_proof wall-time (s): RSA k=17 / SHA256 k=19 / agg k=22 at 1/2/4/8 GPU", "value": res["value"], "unit": "s",
                  "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["ms_per_step"], "higher_is_better": False,
                  "scaling": chain_scaling, "vs_baseline": None, "dtype": "u256 (BN254 Fr/Fq, Montgomery)", "data": "synthetic", "proofs_per_step": 5,
                  "config": {"workload": res["workload"], "workload_short": f"chain: 2 x RSA k=17 + 2 x SHA256-shaped k=19 leaf proofs (Poseidon), barrier, agg k={args.agg_k} proof (Keccak)",
                             "parallelism": res["parallelism"], "k": args.agg_k, "transcript": "evm"}, "proof_bytes": res["proof_bytes"], "proof_sha256": res["proof_sha256"],
                  "comm": comm_fields(res["bytes_gathered_per_step"], "points" if args.shard == "auto" else args.shard, res["collectives_per_step"]),
                  "roofline": res["roofline"], "cpu_baseline": cb, "gpu_proofs": gpu_proofs, "parity": parity, "build": bh,
                  "phase_ms_per_step": res["phase_ms_per_step"], "leaf_proof_sha256": res["leaf_proof_sha256"], "leaf_groups": res["leaf_groups"],
                  **({"replay": replay_block(res["replay_exchanges_per_step"], res.get("replay_trace"))} if replay else {})}, args)
            if parity and parity["bytes_equal"] is False:
                print("bench.py: PARITY FAILURE: a HIP-path proof differs from the CPU oracle's: " + json.dumps([r for r in parity["compared"] if not r["equal"]]), file=sys.stderr, flush=True)
                sys.exit(3)
        teardown()
        return

    out_configs = {}
    head, head_shape = run_config(args.config, args.steps, args.warmup)
    out_configs[args.config] = head
    if world == 1 and not args.no_other_configs:
        for name in ("rsa17", "sha19", "agg22"):
            if name == args.config:
                continue
            try:
                out_configs[name], _ = run_config(name, args.other_steps, 2)
            except Exception as e:   # noqa: BLE001 — never allowed to break the headline measurement
                out_configs[name] = dict(error=str(e)[:300])
        # the k = 22 circuit once more with SURVEY.md 8(d)'s value mix on the free witness cells (50 % 88-bit CRT limbs, 10 % bits, 40 %
        # uniform) instead of uniform field elements: what a real aggregation witness looks like to the commitments (informational)
        try:
            r_, _ = run_config("agg22", max(3, args.other_steps // 2), 1, breakdown_passes=1, with_h2d=False, witness="survey")
            out_configs["agg22_survey_witness"] = {k_: r_[k_] for k_ in ("value", "unit", "steps", "warmup", "ms_per_step", "workload", "witness", "transcript", "kernels_ms_per_step")}
        except Exception as e:   # noqa: BLE001
            out_configs["agg22_survey_witness"] = dict(error=str(e)[:300])

    if world == 1 and not args.no_other_configs and not args.no_chain:
        try:
            out_configs["chain"] = run_chain(max(2, args.other_steps // 3), 1)
        except Exception as e:   # noqa: BLE001
            out_configs["chain"] = dict(error=str(e)[:300])

    # the boundary at each patch level (host-pointer entry points with pageable arrays): the headline configuration and RSA k = 17
    ffi_lv = None
    if world == 1 and not replay and args.ffi_level != "none" and not args.python_schedule and not args.no_other_configs:      # (--no-other-configs: the headline's kernels only — what the profiling passes trace)
        try:
            ffi_lv = ffi_levels(args.config, head["value"])
            if not args.no_other_configs and "rsa17" in out_configs and args.config != "rsa17" and "error" not in out_configs["rsa17"]:
                out_configs["rsa17"]["ffi_levels"] = ffi_levels("rsa17", out_configs["rsa17"]["value"], reps=5)
        except Exception as e:   # noqa: BLE001 — never allowed to break the headline measurement
            import traceback

            traceback.print_exc()
            ffi_lv = dict(error=f"{type(e).__name__}: {e}"[:300])

    # the headline shape at the size the CPU leg's bounded sample runs at: a byte-compared full-shape proof in the default line
    sample = None
    k_cpu = max(19, min(head["k"], args.cpu_baseline_k))
    if args.config == "agg22" and head["k"] > 18 and k_cpu != head["k"] and not args.no_cpu_baseline and not args.replicas:
        try:
            sample = parity_sample("agg22", k_cpu)
        except SystemExit:
            raise
        except Exception as e:   # noqa: BLE001
            if world > 1:
                raise
            sample = dict(error=str(e)[:300])
    if rank == 0:
        dom = head["rooflines"]["msm_accum_affine"]
        out = {
            "metric": "create_proof wall-time (s): RSA k=17 / SHA256 k=19 / agg k=22 at 1/2/4/8 GPU",
            "value": head["value"], "unit": "s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
            "higher_is_better": False, "scaling": ("weak" if args.replicas else "strong") if world == 1 and not replay else (None if replay else ("strong" if shard else "weak")), "vs_baseline": None,
            "proofs_per_step": 1 if (shard or world == 1) else world,
            "dtype": "u256 (BN254 Fr/Fq, Montgomery, 9 x 29-bit limbs in registers / 8 x u32 in HBM)", "data": "synthetic",
            "config": {"workload": head["workload"], "workload_short": head["workload_short"], "headline": args.config, "k": head["k"], "advice": head["advice"], "fixed": head["fixed"],
                       "lookups": head["lookups"], "perm_columns": head["perm_columns"], "degree": head["degree"], "transcript": head["transcript"],
                       "host": "prover.py (Python schedule over the C ABI)" if args.python_schedule else "zkhip_create_proof_ex (schedule and transcript in the library)",
                       "parallelism_short": (f"rank {args.replay_rank} of {args.of}, replayed alone on 1 GPU" if replay else "1 GPU") if world == 1 else (
                           f"one proof sharded x{world}: MSMs by {'point range' if head.get('msm_shard') == 'points' else 'column'}, NTTs by polynomial, sweep by rows" if shard else f"{world} independent proofs, one per GPU, no collective"),
                       "parallelism": (f"rank {args.replay_rank} of {args.of}, replayed alone on 1 GPU" if replay else "1 GPU") if world == 1 else (f"one proof sharded x{world}: MSMs by {'point range (window tables 1/' + str(world) + ' per rank)' if head.get('msm_shard') == 'points' else 'column (whole tables on every rank)'}, coset NTTs by polynomial, then an all-to-all of row windows (own row range + halo of every coset block: 1/N of an all-gather of complete columns), sweep by row range per coset block, the quotient's pieces / h(X) / SHPLONK on row ranges (numerator blocks to their owners and back, Kate division with carries across ranks), evaluations by query, all-gather of the 96-byte partial sums — all inside the library (ncclSend/ncclRecv groups, ncclAllGather)" if shard
                                                                   else f"{world} independent proofs, one per GPU, no collective")},
            "roofline": {k_: dom[k_] for k_ in ("kernel", "bound", "hbm_frac", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms", "note")},
            "int_roofline": dict(kernel="k_accum_affine", **dom["int_roofline"]),
            "configs": out_configs, "build": bh,
            "comm": head["comm"],
            "setup_s": head["setup_s"], "first_proof_s": head["first_proof_s"], "resident_bytes": head["resident_bytes"], "with_h2d": head["with_h2d"],
        }
        out["parity_sample"] = sample
        out["ffi_levels"] = ffi_lv
        if replay:
            out["replay"] = replay_block(head.get("replay_exchanges_per_step"), head.get("replay_trace"))
        out["gpu_proofs"] = gpu_proofs
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"], rsa_cb = cpu_baseline(pv, args, args.config, head["k"], head["transcript"])
                if rsa_cb and "rsa17" in out_configs and "error" not in out_configs["rsa17"]:
                    out_configs["rsa17"]["cpu_baseline"] = rsa_cb
                finish_parity(out)
                cb_, fl_ = out["cpu_baseline"], out.get("ffi_levels")
                if isinstance(fl_, dict) and isinstance(cb_.get("split_s"), dict) and "error" not in fl_:
                    # What a proof would still take at the host-pointer patch levels: the level's GPU-side calls + the CPU remainder (everything that is neither a commitment nor
                    # a transform), taken from THIS run's CPU oracle pass and carried to the headline size like `value`.  An ESTIMATE: the remainder is the oracle's C code on
                    # these host cores, not upstream's rayon code.
                    rest_s = cb_["split_s"]["rest_fraction"] * cb_["value"]
                    for lv in ("curves", "domain"):
                        if isinstance(fl_.get(lv), dict) and "value" in fl_[lv]:
                            fl_[lv]["with_cpu_remainder_estimate_s"] = round(fl_[lv]["value"] + rest_s, 3)
                    fl_["cpu_remainder_estimate"] = dict(value=round(rest_s, 3), unit="s", from_k=cb_["split_s"]["k"], rest_fraction=cb_["split_s"]["rest_fraction"],
                                                         note="the CPU oracle's time outside best_multiexp and the transforms (sweep, permute, grand products, evaluations, SHPLONK arithmetic, "
                                                              "transcript), as a fraction of its pass at the measured size, times cpu_baseline.value")
            except Exception as e:   # noqa: BLE001 — the GPU measurement stands on its own: the line still comes out, the CPU leg says why it is missing
                import traceback

                traceback.print_exc()
                out["cpu_baseline"] = dict(error=f"{type(e).__name__}: {e}"[:300])
                out["parity"] = None
        else:
            out["cpu_baseline"] = None     # N > 1: the GPU-free supervisor of rank 0 times it once the workers are gone (supervise())
            out["parity"] = None           # ... and compares the digests in gpu_proofs with the CPU oracle's
        emit(out, args)
        if out["parity"] and out["parity"]["bytes_equal"] is False:
            print("bench.py: PARITY FAILURE: a HIP-path proof differs from the CPU oracle's: " + json.dumps([r for r in out["parity"]["compared"] if not r["equal"]]), file=sys.stderr, flush=True)
            teardown()
            sys.exit(3)
    teardown()


if __name__ == "__main__":
    main()
