#!/usr/bin/env python3
"""The four spectrum kinds of the reference-layout STFT against T (1024 clips, W = 2048, hop 1024), device resident."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zaf-python_amd"))
import zafx  # noqa: E402

ham = zafx.hamming(2048)
B = 1024
for N in (441000, 441000 + 1024, 441000 + 2048, 441000 + 16 * 1024):
    base = np.stack([np.random.default_rng([0, c]).standard_normal(N).astype(np.float32) for c in range(8)])
    d_x = zafx.DeviceBuffer.from_host(np.tile(base, (B // 8, 1)))
    row = []
    for kind in (False, True, "magnitude", "power"):
        plan = zafx.stft_plan(ham, 1024, onesided=kind)
        d_o = zafx.DeviceBuffer(plan.out_shape(B, N), plan.out_dtype)
        for _ in range(200):
            plan.execute(d_x, d_o, B, N)
        plan.sync()
        plan.timer_start()
        for _ in range(50):
            plan.execute(d_x, d_o, B, N)
        ms = plan.timer_stop() / 50
        row.append(f"{kind}: {ms:.3f} ms {(B * 4 * N + d_o.nbytes) / ms / 1e6:.0f} GB/s")
        d_o.free()
    print("T", plan.out_dims(N)[1], "|", " | ".join(row), flush=True)
    d_x.free()
