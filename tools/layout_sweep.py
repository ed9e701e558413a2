#!/usr/bin/env python3
"""Both 2-D layouts of every transform at W = 2048 (256 clips x 10 s, device resident): ms per launch."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "zaf-python_amd"))
import zafx  # noqa: E402

B, n = 256, 441000
ham, kbd = zafx.hamming(2048), zafx.kaiser_bessel_derived(2048)
fb = zafx.melfilterbank(44100, 2048, 128)
ck = zafx.cqtkernel(44100, 24, 55, 3520)
x = np.random.default_rng(0).standard_normal((8, n)).astype(np.float32)
d_x = zafx.DeviceBuffer.from_host(np.tile(x, (B // 8, 1)))


def timed(pl, a, b, m, reps=10):
    pl.execute(a, b, B, m)
    pl.sync()
    pl.timer_start()
    for _ in range(reps):
        pl.execute(a, b, B, m)
    return pl.timer_stop() / reps


for layout in ("FT", "TF"):
    row = [layout]
    for name, fwd, inv in (
            ("stft", zafx.stft_plan(ham, 1024, layout=layout), zafx.istft_plan(ham, 1024, layout=layout)),
            ("mdct", zafx.mdct_plan(kbd, layout=layout), zafx.mdct_plan(kbd, layout=layout, inverse=True)),
            ("mel", zafx.mel_plan(ham, 1024, fb, layout=layout), None),
            ("mfcc", zafx.mel_plan(ham, 1024, fb, 20, layout=layout), None),
            ("cqt", zafx.cqt_plan(44100, 25, ck, layout=layout), None),
            ("chroma", zafx.cqt_plan(44100, 25, ck, 24, layout=layout), None)):
        T = fwd.out_dims(n)[1]
        d_s = zafx.DeviceBuffer(fwd.out_shape(B, n), fwd.out_dtype)
        txt = f"{name} {timed(fwd, d_x, d_s, n):.3f} ({fwd.last_kernel})"
        if inv is not None:
            d_y = zafx.DeviceBuffer(inv.out_shape(B, T), inv.out_dtype)
            txt += f" inv {timed(inv, d_s, d_y, T):.3f} ({inv.last_kernel})"
            d_y.free()
        d_s.free()
        row.append(txt)
    print(" | ".join(row), flush=True)
