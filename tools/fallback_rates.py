#!/usr/bin/env python3
"""Rates of the float64 fallback paths the host layer routes to (64 clips x 10 s, device resident): ms per launch."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "zaf-python_amd"))
import zafx  # noqa: E402

B, n = 64, 441000
x = np.random.default_rng(0).standard_normal((8, n))
d_x = zafx.DeviceBuffer.from_host(np.tile(x, (B // 8, 1)))


def timed(pl, a, b, m, reps=3):
    pl.execute(a, b, B, m)
    pl.sync()
    pl.timer_start()
    for _ in range(reps):
        pl.execute(a, b, B, m)
    return pl.timer_stop() / reps


def ham(w):
    return 0.54 - 0.46 * np.cos(2 * np.pi * np.arange(w) / w)


for name, fwd, inv in (
        ("stft W=1764 hop=441 (Bluestein)", zafx.stft_plan(ham(1764), 441), zafx.istft_plan(ham(1764), 441)),
        ("stft W=1000 hop=500 (Bluestein)", zafx.stft_plan(ham(1000), 500), zafx.istft_plan(ham(1000), 500)),
        ("mdct W=1920 (Bluestein)", zafx.mdct_plan(np.sin(np.pi / 1920 * (np.arange(1920) + 0.5))),
         zafx.mdct_plan(np.sin(np.pi / 1920 * (np.arange(1920) + 0.5)), inverse=True)),
        ("mel W=4096 hop=2048 (f64 kernel)", zafx.mel_plan(ham(4096), 2048, zafx.melfilterbank(44100, 4096, 128)), None),
        ("istft W=2048 hop=100 (f64 kernels)", zafx.stft_plan(ham(2048), 100, f64=True), zafx.istft_plan(ham(2048), 100))):
    T = fwd.out_dims(n)[1]
    d_s = zafx.DeviceBuffer(fwd.out_shape(B, n), fwd.out_dtype)
    txt = f"{name}: forward {timed(fwd, d_x, d_s, n):8.3f} ms ({fwd.last_kernel})"
    if inv is not None:
        d_y = zafx.DeviceBuffer(inv.out_shape(B, T), inv.out_dtype)
        txt += f" | inverse {timed(inv, d_s, d_y, T):8.3f} ms ({inv.last_kernel})"
        d_y.free()
    print(txt, flush=True)
    d_s.free()
