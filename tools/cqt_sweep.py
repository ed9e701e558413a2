#!/usr/bin/env python3
"""k_cqt at every fft_length (128 clips x 10 s, 24 bins per octave from 55 Hz, 25 frames/s): ms per launch and ns per frame."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "zaf-python_amd"))
import zafx  # noqa: E402

B = 128
for fs in (44100, 22050, 11025, 8000, 4000, 2000):
    n = fs * 10
    ck = zafx.cqtkernel(fs, 24, 55, min(3520, fs / 2.5))
    plan = zafx.cqt_plan(fs, 25, ck)
    x = np.random.default_rng(0).standard_normal((8, n)).astype(np.float32)
    d_x = zafx.DeviceBuffer.from_host(np.tile(x, (B // 8, 1)))
    d_o = zafx.DeviceBuffer(plan.out_shape(B, n), plan.out_dtype)
    plan.execute(d_x, d_o, B, n)
    plan.sync()
    plan.timer_start()
    for _ in range(5):
        plan.execute(d_x, d_o, B, n)
    ms = plan.timer_stop() / 5
    T = plan.out_dims(n)[1]
    print(f"fs {fs:6d}  fft_length {ck.shape[1]:6d}  bins {ck.shape[0]:4d}  nnz {ck.nnz:6d}  T {T:4d}  {ms:8.3f} ms  {ms * 1e6 / (B * T):8.1f} ns/frame  ({plan.last_kernel})", flush=True)
    d_x.free(); d_o.free()
