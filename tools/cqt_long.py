#!/usr/bin/env python3
"""cqtspectrogram at fft_length 65536 (minimum frequency 27.5 Hz): the float32 double form of round 3 against the float64 kernel."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zaf-python_amd"))
import zafx  # noqa: E402

fs, B, N = 44100, 128, 1323000
ck = zafx.cqtkernel(fs, 24, 27.5, 3520)
print("kernel", ck.shape, ck.nnz)
base = np.stack([np.random.default_rng([0, c]).standard_normal(N).astype(np.float32) for c in range(8)])
for f64 in (False, True):
    x = np.tile(base, (B // 8, 1)).astype(np.float64 if f64 else np.float32)
    d_x = zafx.DeviceBuffer.from_host(x)
    plan = zafx.cqt_plan(fs, 25, ck, f64=f64)
    d_o = zafx.DeviceBuffer(plan.out_shape(B, N), plan.out_dtype)
    for _ in range(3):
        plan.execute(d_x, d_o, B, N)
    plan.sync()
    plan.timer_start()
    for _ in range(5):
        plan.execute(d_x, d_o, B, N)
    ms = plan.timer_stop() / 5
    print("f64" if f64 else "f32", plan.last_kernel, f"{ms:.2f} ms for {B} clips x 30 s = {B * N / ms / 1e3:.0f} Msamples/s", flush=True)
    d_x.free(); d_o.free()
