#!/usr/bin/env python3
"""Throughput of the W = 2048 transforms against the frame count T (row alignment of the (F, T) layout); 256 clips."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "zaf-python_amd"))
import zafx  # noqa: E402

B = 256
ham = zafx.hamming(2048)
mel = None
for n in (441000, 441000 + 1024, 441000 + 2 * 1024, 441000 + 4 * 1024, 441000 + 8 * 1024, 441000 + 16 * 1024):
    x = np.random.default_rng(0).standard_normal((8, n)).astype(np.float32)
    d_x = zafx.DeviceBuffer.from_host(np.tile(x, (B // 8, 1)))
    row = []
    for layout, align in (("FT", 0), ("FT", 16), ("TF", 0)):
        for one in (False, True):
            fwd = zafx.stft_plan(ham, 1024, layout=layout, onesided=one, row_align=align)
            inv = zafx.istft_plan(ham, 1024, layout=layout, onesided=one, row_align=align)
            T = fwd.out_dims(n)[1]
            d_s = zafx.DeviceBuffer(fwd.out_shape(B, n), fwd.out_dtype)
            d_y = zafx.DeviceBuffer(inv.out_shape(B, T), inv.out_dtype)
            res = []
            for pl, a, b, m in ((fwd, d_x, d_s, n), (inv, d_s, d_y, T)):
                pl.execute(a, b, B, m)
                pl.sync()
                pl.timer_start()
                for _ in range(10):
                    pl.execute(a, b, B, m)
                res.append(pl.timer_stop() / 10)
            row.append(f"{layout}{'p' if align else ''}{'1' if one else '2'} stft {res[0]:.3f} istft {res[1]:.3f}")
            d_s.free(); d_y.free()
    print(f"T={T:4d} (T%16={T % 16:2d}) | " + " | ".join(row), flush=True)
    d_x.free()

# the real-valued (F, T) outputs: rows are 4 T bytes, 128-B aligned when T % 32 == 0
kbd = zafx.kaiser_bessel_derived(2048)
fb = zafx.melfilterbank(44100, 2048, 128)
for n in (441000 - 1024, 441000, 441000 + 1024, 441000 + 8 * 1024, 441000 + 16 * 1024):
    x = np.random.default_rng(0).standard_normal((8, n)).astype(np.float32)
    d_x = zafx.DeviceBuffer.from_host(np.tile(x, (B // 8, 1)))
    row = []
    for name, fwd, inv in (("mdct", zafx.mdct_plan(kbd), zafx.mdct_plan(kbd, inverse=True)),
                           ("mdct-p", zafx.mdct_plan(kbd, row_align=32), zafx.mdct_plan(kbd, inverse=True, row_align=32)),
                           ("mel", zafx.mel_plan(ham, 1024, fb), None), ("mel-p", zafx.mel_plan(ham, 1024, fb, row_align=32), None)):
        T = fwd.out_dims(n)[1]
        d_s = zafx.DeviceBuffer(fwd.out_shape(B, n), fwd.out_dtype)
        steps = [(fwd, d_x, d_s, n)]
        if inv is not None:
            d_y = zafx.DeviceBuffer(inv.out_shape(B, T), inv.out_dtype)
            steps.append((inv, d_s, d_y, T))
        res = []
        for pl, a, b, m in steps:
            pl.execute(a, b, B, m)
            pl.sync()
            pl.timer_start()
            for _ in range(10):
                pl.execute(a, b, B, m)
            res.append(pl.timer_stop() / 10)
        row.append(f"{name} T={T} (T%32={T % 32}) " + " ".join(f"{r:.3f}" for r in res))
    print(" | ".join(row), flush=True)
    d_x.free()
