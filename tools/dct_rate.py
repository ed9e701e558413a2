#!/usr/bin/env python3
"""Device-resident rate of zaf.dct / dst of one length and type.  usage: dct_rate.py N type [sine] [rows]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "zaf-python_amd"))
import zafx  # noqa: E402

n, t = int(sys.argv[1]), int(sys.argv[2])
sine = len(sys.argv) > 3 and sys.argv[3] == "1"
rows = int(sys.argv[4]) if len(sys.argv) > 4 else 16384
pl = zafx.dct_plan(n, t, sine)
d_x = zafx.DeviceBuffer.from_host(np.random.default_rng(0).standard_normal((rows, n)).astype(np.float32))
d_y = zafx.DeviceBuffer((rows, n), np.float32)
pl.execute(d_x, d_y, rows, n)
pl.sync()
pl.timer_start()
for _ in range(20):
    pl.execute(d_x, d_y, rows, n)
ms = pl.timer_stop() / 20
print(f"{'dst' if sine else 'dct'}{t} N={n} x {rows}: {ms:.4f} ms = {rows * n * 8 / ms / 1e9:.3f} TB/s = {rows * n * 8 / ms / 8e9:.3f} of HBM ({pl.last_kernel})", flush=True)
