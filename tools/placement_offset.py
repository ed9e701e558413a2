#!/usr/bin/env python3
"""Does the headline's rate follow the OFFSET of its output inside one allocation (an aliasing of the write pattern on the channel hash) or only the allocation?
One allocation with 512 MiB of slack, the output placed at a sweep of offsets inside it.  usage: placement_offset.py"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "zaf-python_amd"))
import zafx  # noqa: E402

B, N, W, H = 1024, 441000, 2048, 1024
x = np.random.default_rng(0).standard_normal((8, N)).astype(np.float32)
d_x = zafx.DeviceBuffer.from_host(np.tile(x, (B // 8, 1)))
pl = zafx.stft_plan(zafx.hamming(W), H)
shape = pl.out_shape(B, N)
nbytes = int(np.prod(shape)) * 8
big = zafx.DeviceBuffer((nbytes + (512 << 20),), np.uint8)
base = big.ptr.value if hasattr(big.ptr, "value") else int(big.ptr)


class Raw:
    def __init__(self, ptr):
        self.ptr = ctypes.c_void_p(ptr)


for off in (0, 4 << 10, 64 << 10, 1 << 20, 2 << 20, 6 << 20, 16 << 20, 27 << 20, 64 << 20, 128 << 20, 256 << 20, 511 << 20, 0):
    d = Raw(base + off)
    for _ in range(100):
        pl.execute(d_x, d, B, N)
    pl.sync()
    pl.timer_start()
    for _ in range(30):
        pl.execute(d_x, d, B, N)
    print(f"offset {off / 2**20:9.3f} MiB: {pl.timer_stop() / 30:.4f} ms", flush=True)
