"""CPU: libzkhip.so loads and exports every symbol include/zkhip.h declares (no compute without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g

    g.build()
    import halo2_zkcert_amd.ffi as ffi

    return ffi


def test_header_symbols_exported(built):
    ffi = built
    hdr = open(os.path.join(ROOT, "include", "zkhip.h")).read()
    declared = sorted(set(re.findall(r"\b(zkhip_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    L = ctypes.CDLL(ffi.LIB_PATH)
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, f"declared in zkhip.h but not exported: {missing}"
    assert sorted(ffi.SYMBOLS) == declared


def test_fails_loudly_without_gpu(built):
    ffi = built
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ffi.ZkhipError):
        ffi.Context(0)
    h = ctypes.c_void_p()
    rc = ffi.lib().zkhip_init(ctypes.byref(h), 0)
    assert rc != 0 and ffi.lib().zkhip_last_error()


def test_host_point_helpers(built, oracle):
    """zkhip_g1_to_affine / zkhip_g1_to_bytes are host code in the product library: check vs golden."""
    ffi = built
    zo = oracle
    from util import H, load

    g = load("g1.json")
    for m in g["mul_gen"][:6]:
        aff = zo.affine_from_ints([(H(m["x"]), H(m["y"]))])[0]
        assert ffi.g1_to_bytes(aff).hex() == m["compressed"]
        # scale to a non-trivial Jacobian representative (x z^2, y z^3, z) and normalise back
        z = zo.fq_from_int(0x1234567)
        z2 = zo._binary("zko_fq_mul", z, z)
        z3 = zo._binary("zko_fq_mul", z2, z)
        import numpy as np

        jac = np.concatenate([zo._binary("zko_fq_mul", aff[:4], z2), zo._binary("zko_fq_mul", aff[4:], z3), z])
        assert (ffi.g1_to_affine(jac) == aff).all()
    import numpy as np

    ident = np.zeros(12, dtype=np.uint64)
    assert (ffi.g1_to_affine(ident) == 0).all()
    assert ffi.g1_to_bytes(np.zeros(8, dtype=np.uint64)).hex() == g["identity_compressed"]


def test_missing_library_is_an_error_not_a_fallback(built, monkeypatch, tmp_path):
    ffi = built
    monkeypatch.setattr(ffi, "_LIB", None)
    monkeypatch.setattr(ffi, "LIB_PATH", str(tmp_path / "libzkhip.so"))
    with pytest.raises(ffi.ZkhipError, match="no CPU fallback"):
        ffi.lib()


def test_rust_bindings_are_generated_from_the_header(built):
    """integration/rust/zkhip-sys/src/lib.rs is tools/gen_rust_ffi.py's output for the current include/zkhip.h and declares every
    exported function; the #[repr(C)] structs list the same fields in the same order as the ctypes structures the tests run on."""
    import subprocess
    import sys

    ffi = built
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_ffi.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    rs = open(os.path.join(ROOT, "integration", "rust", "zkhip-sys", "src", "lib.rs")).read()
    for s in ffi.SYMBOLS:
        assert re.search(rf"\bpub fn {s}\(", rs), s
    for rust_name, cls in (("zk_graph", ffi.ZkGraph), ("zk_evalh_args", ffi.ZkEvalhArgs), ("zk_transcript", ffi.ZkTranscript),
                           ("zk_proving_key", ffi.ZkProvingKey), ("zk_proof_out", ffi.ZkProofOut), ("zk_blinding", ffi.ZkBlinding),
                           ("zk_proof_inputs", ffi.ZkProofInputs)):
        body = re.search(rf"pub struct {rust_name} \{{(.*?)\n\}}", rs, flags=re.S).group(1)
        fields = re.findall(r"pub (\w+):", body)
        assert fields == [f[0] for f in cls._fields_], rust_name
    # and the same field lists appear, in order, in the C header
    hdr = open(os.path.join(ROOT, "include", "zkhip.h")).read()
    for cname, cls in (("zk_proof_inputs", ffi.ZkProofInputs), ("zk_blinding", ffi.ZkBlinding), ("zk_proof_out", ffi.ZkProofOut)):
        body = re.search(rf"typedef struct {cname} \{{(.*?)\}} {cname};", hdr, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = re.findall(r"(\w+)\s*(?:\[\d*\])?\s*[;,]", body)
        assert names == [f[0] for f in cls._fields_], cname


def test_bench_build_hash_is_the_device_code():
    """bench.build_hash: sha256 of libzkhip.so's .hip_fatbin section (what keys the committed counter passes under profiles/); it must
    parse the shipped library and match the `# build=` header of the newest pass, or roofline.traffic would silently become null"""
    import glob
    import importlib

    bench = importlib.import_module("bench")
    bh = bench.build_hash()
    assert len(bh) == 16 and int(bh, 16) >= 0
    heads = {open(f).readline().strip() for f in glob.glob(os.path.join(ROOT, "profiles", "*_pmc_agg22.csv"))}
    if "# build=" + bh not in heads:
        import pytest

        pytest.skip(f"no counter pass under profiles/ was taken on device code {bh}: run tools/profile_round.sh (roofline.traffic is null until then)")
    assert bench.pmc_traffic("agg22", bh)[0]["msm_accum_affine"] > 1e9
