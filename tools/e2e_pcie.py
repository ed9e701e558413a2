#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer API (never the bench `value`): host f32 in -> device -> host c64 out."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "zaf-python_amd"))
import zafx  # noqa: E402

B, N, W, H = 128, 441000, 2048, 1024
x = np.random.default_rng(0).standard_normal((B, N)).astype(np.float32)
w = zafx.hamming(W)
for onesided in (False, True):
    zafx.stft_batch(x[:4], w, H, onesided=onesided)   # plan + first-touch
    t0 = time.perf_counter()
    out = zafx.stft_batch(x, w, H, onesided=onesided)
    dt = time.perf_counter() - t0
    plan = zafx.stft_plan(w, H, onesided=onesided)
    d_in = zafx.DeviceBuffer.from_host(x)
    d_out = zafx.DeviceBuffer(plan.out_shape(B, N), plan.out_dtype)
    t1 = time.perf_counter()
    d_in.upload(x)
    t2 = time.perf_counter()
    plan.execute(d_in, d_out, B, N)
    plan.sync()
    t3 = time.perf_counter()
    res = d_out.download()
    t4 = time.perf_counter()
    host = zafx.pinned_empty(d_out.shape, d_out.dtype)   # allocated once, reused by a real caller
    d_out.download(out=host)                             # first touch
    t5 = time.perf_counter()
    d_out.download(out=host)
    t6 = time.perf_counter()
    print(f"onesided={onesided}: stft_batch {B} clips {dt * 1e3:.1f} ms = {B * N / dt / 1e6:.0f} Msamples/s | "
          f"H2D {x.nbytes / 1e6:.0f} MB {(t2 - t1) * 1e3:.1f} ms ({x.nbytes / (t2 - t1) / 1e9:.1f} GB/s) | kernel {(t3 - t2) * 1e3:.2f} ms | "
          f"D2H {res.nbytes / 1e6:.0f} MB pageable {(t4 - t3) * 1e3:.1f} ms ({res.nbytes / (t4 - t3) / 1e9:.1f} GB/s), "
          f"pinned+reused {(t6 - t5) * 1e3:.1f} ms ({res.nbytes / (t6 - t5) / 1e9:.1f} GB/s)")
