#!/usr/bin/env python3
"""Device-resident throughput of the transforms at other window sizes (sanity sweep; 256 clips x 10 s)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "zaf-python_amd"))
import zafx  # noqa: E402

B, N = 256, 441000
x = np.random.default_rng(0).standard_normal((8, N)).astype(np.float32)
d_x = zafx.DeviceBuffer.from_host(np.tile(x, (B // 8, 1)))


def run(plan, d_in, n_in, reps=10):
    d_out = zafx.DeviceBuffer(plan.out_shape(B, n_in), plan.out_dtype)
    plan.execute(d_in, d_out, B, n_in)
    plan.sync()
    plan.timer_start()
    for _ in range(reps):
        plan.execute(d_in, d_out, B, n_in)
    ms = plan.timer_stop() / reps
    return ms, d_out


for wl in (128, 256, 512, 1024, 2048, 4096, 8192):
    ham, kbd = zafx.hamming(wl), zafx.kaiser_bessel_derived(wl)
    row = [f"W={wl:5d}"]
    for layout in ("FT", "TF"):
        fwd = zafx.stft_plan(ham, wl // 2, layout=layout)
        ms, d_s = run(fwd, d_x, N)
        T = fwd.out_dims(N)[1]
        ims, _ = run(zafx.istft_plan(ham, wl // 2, layout=layout), d_s, T)
        gb = B * (4 * N + 8 * wl * T) / 1e9
        row.append(f"stft[{layout}] {ms:6.3f} ms {gb / ms:5.2f} TB/s {fwd.last_kernel:12s} istft {ims:6.3f} ms {gb / ims:5.2f} TB/s")
        d_s.free()
    m = zafx.mdct_plan(kbd)
    ms, d_m = run(m, d_x, N)
    T = m.out_dims(N)[1]
    ims, _ = run(zafx.mdct_plan(kbd, inverse=True), d_m, T)
    gb = B * (4 * N + 4 * (wl // 2) * T) / 1e9
    row.append(f"mdct {ms:6.3f} ms {gb / ms:5.2f} TB/s imdct {ims:6.3f} ms {gb / ims:5.2f} TB/s")
    print(" | ".join(row), flush=True)
