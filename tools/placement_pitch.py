#!/usr/bin/env python3
"""Is the placement effect an aliasing of the clip stride?  The headline STFT onto arrays of row pitch 432 (compact), 448, 512 -- clip strides 6.75, 7, 8 MiB --
each in three successive allocations of one process (hipMalloc: ZAFX_ALLOC_CHUNK_MB=0).  ms per launch over the SAME algorithmic bytes."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "zaf-python_amd"))
import zafx  # noqa: E402

B, N, W, H = 1024, 441000, 2048, 1024
x = np.random.default_rng(0).standard_normal((8, N)).astype(np.float32)
d_x = zafx.DeviceBuffer.from_host(np.tile(x, (B // 8, 1)))
keep = []
for rnd in range(3):
    for align in (0, 64, 512, 16):
        pl = zafx.stft_plan(zafx.hamming(W), H, row_align=align)
        d = zafx.DeviceBuffer(pl.out_shape(B, N), pl.out_dtype)
        for _ in range(100):
            pl.execute(d_x, d, B, N)
        pl.sync()
        pl.timer_start()
        for _ in range(30):
            pl.execute(d_x, d, B, N)
        print(f"round {rnd} pitch {pl.out_shape(B, N)[2]}: {pl.timer_stop() / 30:.4f} ms", flush=True)
        keep.append(d)   # (held: the next allocation lands elsewhere)
