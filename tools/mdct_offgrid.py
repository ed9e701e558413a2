#!/usr/bin/env python3
"""k_mdct_ft32 against the frame count T (1024 clips, compact reference layout): the carry form of round 3 (T % 16 != 0)."""
import sys, os, time
sys.path.insert(0, "zaf-python_amd")
import numpy as np, zafx
kbd = zafx.kaiser_bessel_derived(2048)
B = 1024
for N in (441000, 441000 + 1024, 441000 + 2048, 441000 + 4096, 441000 + 8 * 1024):
    base = np.stack([np.random.default_rng([0, c]).standard_normal(N).astype(np.float32) for c in range(8)])
    d_x = zafx.DeviceBuffer.from_host(np.tile(base, (B // 8, 1)))
    plan = zafx.mdct_plan(kbd)
    d_o = zafx.DeviceBuffer(plan.out_shape(B, N), plan.out_dtype)
    for _ in range(300): plan.execute(d_x, d_o, B, N)
    plan.sync()
    plan.timer_start()
    for _ in range(50): plan.execute(d_x, d_o, B, N)
    ms = plan.timer_stop() / 50
    print(os.environ.get("ZAFX_LIBRARY", "default")[-18:], "T", plan.out_dims(N)[1], round(ms, 4), "ms", round((B * 4 * N + d_o.nbytes) / ms / 1e6), "GB/s", flush=True)
    d_x.free(); d_o.free()
