#!/usr/bin/env python3
"""Per-phase cycle counters of k_istft_ft16 (needs a library built with -DZAFX_PROF; see profiles/r01_notes.md).

    ZAFX_LIBRARY=tools/bin/libzafx_prof.so python tools/prof_istft.py
"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "zaf-python_amd"))
import zafx  # noqa: E402
from zafx import _lib  # noqa: E402

lib = _lib.load()
B, N, W, H = 1024, 441000, 2048, 1024
ham = zafx.hamming(W)
fwd, inv = zafx.stft_plan(ham, H), zafx.istft_plan(ham, H)
x = np.random.default_rng(0).standard_normal((8, N)).astype(np.float32)
d_x = zafx.DeviceBuffer.from_host(np.tile(x, (B // 8, 1)))
F, T = fwd.out_dims(N)
d_s = zafx.DeviceBuffer((B, F, T), np.complex64)
fwd.execute(d_x, d_s, B, N)
fwd.sync()
d_y = zafx.DeviceBuffer((B, inv.out_dims(T)[0]), np.float32)
out = (ctypes.c_ulonglong * 16)()
inv.execute(d_s, d_y, B, T)
inv.sync()
lib.zafx_debug_prof(out)
reps = 5
for _ in range(reps):
    inv.execute(d_s, d_y, B, T)
inv.sync()
lib.zafx_debug_prof(out)
names = ["loop top", "issue A", "FFT", "barrier", "issue B", "OLA+carry", "barrier", "fold A", "fold B(+barrier at top)"]
tiles = reps * 27 * 4
tot = 0
for i, n in enumerate(names):
    print(f"{n:28s} {out[i] / tiles:10.0f} cyc/tile")
    tot += out[i] / tiles
print("total", tot)
