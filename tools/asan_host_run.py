# host-side code of the library under ASan/UBSan (CPU only: transcripts, Poseidon, host point helpers, comm/option plumbing, error paths)
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')]
import halo2_zkcert_amd.ffi as ffi
ffi.LIB_PATH = os.environ['ZKHIP_ASAN_LIB']
import numpy as np, ctypes as C
import zkoracle_py as zo, pyref as P
import halo2_zkcert_amd.prover as pv
zo.build(); zo.lib()
L = ffi.lib()
# transcripts
for kind in ("blake2b", "poseidon", "evm"):
    t = ffi.LibTranscript(kind)
    for i in range(1, 40):
        pt = zo.g1_mul_gen(zo.fr_from_int(i * 7 + 1))
        t.write_point(pt)
        t.common_scalar(zo.fr_from_int(i))
        if i % 3 == 0: t.write_scalar(zo.fr_from_int(pow(i, 33, P.R)))
        if i % 4 == 0: t.squeeze_limbs()
    assert len(t.proof()) > 0 and len(t.challenges()) == 9
    del t
st = np.stack([pv.fr_from_int_host(v) for v in (0, 1, 2)])
assert [pv.from_mont_host(r) for r in ffi.poseidon_permute(st)] == P.POSEIDON_KAT
assert [pv.from_mont_host(r) for r in ffi.poseidon_permute(st, plain=True)] == P.POSEIDON_KAT
# host point helpers
rows = [np.asarray(zo.g1_mul_gen(zo.fr_from_int(k)), dtype=np.uint64).reshape(-1)[:12] for k in (1, 5, 99)]
rows = [r if r.size == 12 else np.concatenate([r, zo.fq_from_int(1)]) for r in rows]
jac = np.stack(rows + [np.array([0]*4 + [int(v) for v in zo.fq_from_int(1)] + [0]*4, dtype=np.uint64)])
aff = ffi.g1_batch_to_affine(jac)
assert (aff[3] == 0).all() and ffi.g1_to_bytes(aff[3])[31] == 0x80
ffi.g1_add(jac[0], jac[1]); ffi.g1_to_affine(jac[2])
ffi.keccak256(b"abc" * 100)
# error paths without a GPU
h = C.c_void_p()
assert L.zkhip_init(C.byref(h), 0) != 0
assert L.zkhip_comm_destroy(None) != 0 and L.zkhip_set_option(None, b"x", 1) != 0
# round 6's entry points: argument validation with no context behind them
out12 = (C.c_uint64 * 12)()
n_ = C.c_size_t()
assert L.zkhip_msm_g1(None, None, None, 0, out12) != 0 and L.zkhip_msm_g1_batch(None, None, None, 0, 0, out12) != 0
assert L.zkhip_comm_trace(None, 1) != 0 and L.zkhip_comm_trace_read(None, 0, C.byref(n_), None, None, None, None, None, None) != 0
assert L.zkhip_comm_phase_name(0) is not None and L.zkhip_comm_phase_name(200) is not None
assert b"null" in L.zkhip_last_error()
buf = (C.c_uint8 * 128)()
L.zkhip_comm_unique_id(buf)   # may fail (no librccl / no device): must not crash
print("asan host run ok")
