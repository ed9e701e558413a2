#!/usr/bin/env python3
"""Device-resident rates of melspectrogram / mfcc against the window (hop W / 2, 128 filters, 20 coefficients), 1024 clips x 10 s."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "zaf-python_amd"))
import zafx  # noqa: E402

B, N = 1024, 441000
x = np.random.default_rng(0).standard_normal((8, N)).astype(np.float32)
d_x = zafx.DeviceBuffer.from_host(np.tile(x, (B // 8, 1)))
for W in (256, 512, 1024, 2048, 4096, 8192):
    fb = zafx.melfilterbank(44100, W, 128 if W >= 1024 else 40)
    for nc in (None, 20):
        pl = zafx.mel_plan(zafx.hamming(W), W // 2, fb, nc)
        d = zafx.DeviceBuffer(pl.out_shape(B, N), pl.out_dtype)
        for _ in range(20):
            pl.execute(d_x, d, B, N)
        pl.sync()
        pl.timer_start()
        for _ in range(10):
            pl.execute(d_x, d, B, N)
        ms = pl.timer_stop() / 10
        print(f"W={W:5d} {'mfcc' if nc else 'mel '}: {ms:7.3f} ms = {B * N / ms / 1e6:7.1f} Gsamples/s ({pl.last_kernel})", flush=True)
        d.free()
