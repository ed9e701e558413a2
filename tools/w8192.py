#!/usr/bin/env python3
"""W = 8192 in the reference layout, device resident, 1024 clips x 10 s: rate of the STFT kinds (k_stft_ft16q: four classes of bins per
16-frame tile) on and off the line grid.  ZAFX_LIBRARY selects a variant build (tools/build_variant.sh).

    python tools/w8192.py [hop ...]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zaf-python_amd"))
import zafx  # noqa: E402

B = 1024
hops = [int(a) for a in sys.argv[1:]] or [4096]
rng = np.random.default_rng(5)
for hop in hops:
    for n in (441000, hop * 111):
        x = rng.standard_normal((8, n)).astype(np.float32)
        d_in = zafx.DeviceBuffer.from_host(np.tile(x, (B // 8, 1)))
        for kw in ({}, {"onesided": True}, {"onesided": "magnitude"}):
            plan = zafx.stft_plan(zafx.hamming(8192), hop, **kw)
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
            print(f"hop {hop} n {n} {str(kw):28s} {plan.last_kernel:14s} T {shape[-1]:4d}: {ms:7.3f} ms  {nbytes / ms / 1e9:6.2f} TB/s", flush=True)
            d_out.free()
        d_in.free()
