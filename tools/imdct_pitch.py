#!/usr/bin/env python3
"""Device-resident rate of the IMDCT against the row pitch of its input.  usage: imdct_pitch.py T row_align [clips [W]]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "zaf-python_amd"))
import zafx  # noqa: E402

T, align = int(sys.argv[1]), int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
W = int(sys.argv[4]) if len(sys.argv) > 4 else 8192
w = zafx.kaiser_bessel_derived(W)
pl = zafx.mdct_plan(w, inverse=True, row_align=align)
pitch = -(-T // max(align, 1)) * max(align, 1)
d_c = zafx.DeviceBuffer.from_host(np.random.default_rng(0).standard_normal((B, W // 2, pitch)).astype(np.float32))
d_y = zafx.DeviceBuffer((B, pl.out_dims(T)[0]), np.float32)
pl.execute(d_c, d_y, B, T)
pl.sync()
pl.timer_start()
for _ in range(10):
    pl.execute(d_c, d_y, B, T)
ms = pl.timer_stop() / 10
gb = B * (4 * (W // 2) * T + 4 * ((W // 2) * (T - 1) - 1)) / 1e9
print(f"imdct W={W} T={T} pitch={pitch}: {ms:.3f} ms = {gb / ms:.2f} TB/s = {gb / ms / 8:.3f} of HBM ({pl.last_kernel})", flush=True)
