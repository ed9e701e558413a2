#!/usr/bin/env python3
"""The headline's output in memory from HIP's virtual-memory API (tools/exp_vmm.hip): physical chunks of a chosen size.  usage: placement_vmm.py [chunk_MiB ...]  (0 = hipMalloc)"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "zaf-python_amd"))
import zafx  # noqa: E402

vmm = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "libvmm.so"))
vmm.vmm_alloc.restype = ctypes.c_void_p
vmm.vmm_alloc.argtypes = [ctypes.c_size_t, ctypes.c_size_t]
vmm.vmm_granularity.restype = ctypes.c_size_t
B, N, W, H = 1024, 441000, 2048, 1024
x = np.random.default_rng(0).standard_normal((8, N)).astype(np.float32)
d_x = zafx.DeviceBuffer.from_host(np.tile(x, (B // 8, 1)))
pl = zafx.stft_plan(zafx.hamming(W), H)
shape = pl.out_shape(B, N)
nbytes = int(np.prod(shape)) * 8
print("granularity min / recommended:", vmm.vmm_granularity(0), vmm.vmm_granularity(1), flush=True)


class Raw:
    def __init__(self, ptr):
        self.ptr = ctypes.c_void_p(ptr)


vmm.vmm_set_align.argtypes = [ctypes.c_size_t]
vmm.vmm_set_align(int(os.environ.get("VMM_ALIGN_MB", "0")) << 20)
for mib in [int(a) for a in sys.argv[1:]] or [0, 2, 64, 1024, 0]:
    if mib == 0:
        d = zafx.DeviceBuffer(shape, pl.out_dtype)
    else:
        p = vmm.vmm_alloc(nbytes, mib << 20)
        if not p:
            print(mib, "MiB chunks: allocation failed", flush=True)
            continue
        d = Raw(p)
    for _ in range(200):
        pl.execute(d_x, d, B, N)
    pl.sync()
    pl.timer_start()
    for _ in range(50):
        pl.execute(d_x, d, B, N)
    print(f"chunk {mib:5d} MiB: {pl.timer_stop() / 50:.4f} ms", flush=True)
