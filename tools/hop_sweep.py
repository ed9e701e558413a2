#!/usr/bin/env python3
"""W = 2048 against the hop (W/2 is what the configs use; W/4 and W/8 are as common in practice): device-resident rates of
stft / istft / mel, 256 clips x 10 s.   gpurun -- 'python tools/hop_sweep.py'"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "zaf-python_amd"))
import zafx  # noqa: E402

B, N, W = 256, 441000, 2048
x = np.random.default_rng(0).standard_normal((8, N)).astype(np.float32)
d_x = zafx.DeviceBuffer.from_host(np.tile(x, (B // 8, 1)))
w = zafx.hamming(W)
fb = zafx.melfilterbank(44100, W, 128)


def timed(plan, d_in, d_out, n_in, reps=20):
    for _ in range(5):
        plan.execute(d_in, d_out, B, n_in)
    plan.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        plan.execute(d_in, d_out, B, n_in)
    plan.sync()
    return (time.perf_counter() - t0) / reps * 1e3


for hop in (1024, 512, 256, 441, 1000):
    fwd = zafx.stft_plan(w, hop)
    F, T = fwd.out_dims(N)
    d_s = zafx.DeviceBuffer((B, F, T), np.complex64)
    ms = timed(fwd, d_x, d_s, N)
    gb = B * (4 * N + 8 * F * T) / 1e9
    line = f"hop {hop:5d} T {T:5d}: stft {ms:7.3f} ms {gb / ms:6.2f} TB/s ({fwd.last_kernel})"
    if hop <= W:
        inv = zafx.istft_plan(w, hop)
        d_y = zafx.DeviceBuffer((B, inv.out_dims(T)[0]), np.float32)
        ms = timed(inv, d_s, d_y, T)
        line += f" | istft {ms:7.3f} ms {gb / ms:6.2f} TB/s ({inv.last_kernel})"
        d_y.free()
    mel = zafx.mel_plan(w, hop, fb)
    d_m = zafx.DeviceBuffer((B,) + tuple(mel.out_dims(N)), np.float32)
    ms = timed(mel, d_x, d_m, N)
    line += f" | mel {ms:7.3f} ms ({B * N / ms / 1e3:8.0f} Msamples/s)"
    print(line)
    d_s.free()
    d_m.free()
