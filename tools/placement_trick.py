#!/usr/bin/env python3
"""Does the headline's output land better when a larger array was allocated and freed first?  usage: placement_trick.py [0|1|2]  (0: plain, 1: 2x dummy freed first, 2: 4x)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "zaf-python_amd"))
import zafx  # noqa: E402

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
B, N, W, H = 1024, 441000, 2048, 1024
x = np.random.default_rng(0).standard_normal((8, N)).astype(np.float32)
d_x = zafx.DeviceBuffer.from_host(np.tile(x, (B // 8, 1)))
pl = zafx.stft_plan(zafx.hamming(W), H)
shape = pl.out_shape(B, N)
if mode:
    dummy = zafx.DeviceBuffer((shape[0] * 2 * mode,) + tuple(shape[1:]), pl.out_dtype)
    dummy.free()
d = zafx.DeviceBuffer(shape, pl.out_dtype)
for _ in range(200):
    pl.execute(d_x, d, B, N)
pl.sync()
pl.timer_start()
for _ in range(50):
    pl.execute(d_x, d, B, N)
print(f"mode {mode}: {pl.timer_stop() / 50:.4f} ms ({pl.last_kernel})", flush=True)
