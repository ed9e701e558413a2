#!/usr/bin/env python3
"""Is a slow placement made of slow CHUNKS?  232 separate physical allocations of 64 MiB (14.5 GB), each filled 20 times: microseconds per fill, then the headline
on the library's own allocation.  usage: placement_chunks.py"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "zaf-python_amd"))
import zafx  # noqa: E402

vmm = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "libvmm.so"))
n = 232
out = (ctypes.c_float * n)()
rc = vmm.vmm_probe(ctypes.c_size_t(64 << 20), n, 20, out)
t = np.array(out[:])
print("probe rc", rc, "us per 64-MiB fill: min %.2f median %.2f max %.2f" % (t.min(), np.median(t), t.max()), "| TB/s median %.2f" % (67.108864 / np.median(t)))
print("histogram (us):", np.histogram(t, bins=8))
print("first 40:", np.round(t[:40], 1))
B, N, W, H = 1024, 441000, 2048, 1024
x = np.random.default_rng(0).standard_normal((8, N)).astype(np.float32)
d_x = zafx.DeviceBuffer.from_host(np.tile(x, (B // 8, 1)))
pl = zafx.stft_plan(zafx.hamming(W), H)
d = zafx.DeviceBuffer(pl.out_shape(B, N), pl.out_dtype)
for _ in range(200):
    pl.execute(d_x, d, B, N)
pl.sync()
pl.timer_start()
for _ in range(50):
    pl.execute(d_x, d, B, N)
print(f"headline on zafx_alloc: {pl.timer_stop() / 50:.4f} ms", flush=True)
