import sys,time; sys.path.insert(0,'.')
import numpy as np, ctypes as C
import halo2_zkcert_amd.ffi as ffi
t=ffi.LibTranscript("poseidon")
s=np.array([5,6,7,8],dtype=np.uint64); sp=s.ctypes.data_as(C.POINTER(C.c_uint64))
cb=t._cb
for rep in range(3):
    t0=time.perf_counter()
    for _ in range(2000): cb.common_scalar(cb.user, sp)
    t1=time.perf_counter()
    out=(C.c_uint64*4)()
    for _ in range(1000): cb.squeeze_challenge(cb.user, out)
    t2=time.perf_counter()
    print("absorb: us per perm", (t1-t0)/1000*1e6, " squeeze: us per perm", (t2-t1)/1000*1e6)
L=ffi.lib(); st=np.zeros((3,4),dtype=np.uint64); st[0,0]=5; p=st.ctypes.data_as(C.c_void_p)
for f in (L.zkhip_poseidon_permute_plain, L.zkhip_poseidon_permute):
    t0=time.perf_counter()
    for _ in range(5000): f(p)
    print("perm us", (time.perf_counter()-t0)/5000*1e6)
