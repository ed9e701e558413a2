#!/usr/bin/env python3
"""W = 8192 MDCT in the reference layout, device resident, 1024 clips: rate on and off the line grid (k_mdct_ft32q).  ZAFX_LIBRARY selects a
variant build (tools/build_variant.sh)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zaf-python_amd"))
import zafx  # noqa: E402

B = 1024
rng = np.random.default_rng(5)
kbd = zafx.kaiser_bessel_derived(8192)
for n in (441000, 4096 * 111, 4096 * 127):
    x = rng.standard_normal((8, n)).astype(np.float32)
    d_in = zafx.DeviceBuffer.from_host(np.tile(x, (B // 8, 1)))
    plan = zafx.mdct_plan(kbd)
    shape = plan.out_shape(B, n)
    d_out = zafx.DeviceBuffer(shape, plan.out_dtype)
    for _ in range(3):
        plan.execute(d_in, d_out, B, n)
    plan.sync()
    plan.timer_start()
    for _ in range(10):
        plan.execute(d_in, d_out, B, n)
    ms = plan.timer_stop() / 10
    nbytes = B * n * 4 + d_out.nbytes
    print(f"n {n} {plan.last_kernel:14s} T {shape[-1]:4d}: {ms:7.3f} ms  {nbytes / ms / 1e9:6.2f} TB/s", flush=True)
    d_out.free()
    d_in.free()
