#!/usr/bin/env python3
"""Device-resident rate of the float64 CQT (clips x 30 s): ms per launch.  usage: cqt64_rate.py [clips]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "zaf-python_amd"))
import zafx  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = 1323000
ck = zafx.cqtkernel(44100, 24, 55, 3520)
x = np.random.default_rng(0).standard_normal((8, n))
pl = zafx.cqt_plan(44100, 25, ck, f64=True)
d_x = zafx.DeviceBuffer.from_host(np.tile(x, (B // 8, 1)))
d_s = zafx.DeviceBuffer(pl.out_shape(B, n), pl.out_dtype)
pl.execute(d_x, d_s, B, n)
pl.sync()
pl.timer_start()
for _ in range(3):
    pl.execute(d_x, d_s, B, n)
ms = pl.timer_stop() / 3
print(f"cqt f64 {os.environ.get('ZAFX_LIBRARY', 'shipped').split('/')[-1]}: {B} clips {ms:.3f} ms = {B * n / ms / 1e6:.1f} Gsamples/s ({pl.last_kernel})", flush=True)
