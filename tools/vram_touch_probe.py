#!/usr/bin/env python3
"""Is the FIRST touch of HBM slow on this box?  (cold-start analysis, VERDICT r3 item 3: the driver's fresh box paid 1.3 s for a first proof
that takes 0.16 s on others.)  Run as the first GPU process of a gpurun call:
    python tools/vram_touch_probe.py [GiB]
Allocates GiB of device memory in 4 GiB blocks straight from hipMalloc (no caching allocator in the way), times the allocation, a first
memset over all of it and a second one."""
import ctypes
import sys
import time

t0 = time.perf_counter()
import torch  # noqa: F401  (binds torch's libamdhip64, the runtime every process of this repo uses)

t_imp = time.perf_counter() - t0
hip = ctypes.CDLL("libamdhip64.so")
gib = int(sys.argv[1]) if len(sys.argv) > 1 else 40
t0 = time.perf_counter()
assert hip.hipInit(0) == 0 and hip.hipSetDevice(0) == 0
hip.hipDeviceSynchronize()
t_init = time.perf_counter() - t0
blocks, blk = [], 4 << 30
t0 = time.perf_counter()
for _ in range(gib // 4):
    p = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(blk)) == 0
    blocks.append(p)
t_alloc = time.perf_counter() - t0


def touch():
    t = time.perf_counter()
    for p in blocks:
        assert hip.hipMemsetAsync(p, 0, ctypes.c_size_t(blk), None) == 0
    assert hip.hipDeviceSynchronize() == 0
    return time.perf_counter() - t


t1, t2, t3 = touch(), touch(), touch()
print(f"  import torch {t_imp:.3f} s; hipInit + first sync {t_init:.3f} s; hipMalloc {gib} GiB in {len(blocks)} blocks {t_alloc * 1e3:.1f} ms; "
      f"memset all: first {t1 * 1e3:.1f} ms ({gib * 1.073741824 / t1:.0f} GB/s), second {t2 * 1e3:.1f} ms, third {t3 * 1e3:.1f} ms")
