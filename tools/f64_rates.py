#!/usr/bin/env python3
"""Device-resident rate of the float64 mode of every transform (128 clips x 10 s, W = 2048; CQT: 32 clips): ms per launch."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "zaf-python_amd"))
import zafx  # noqa: E402

n = 441000
ham, kbd = zafx.hamming(2048), zafx.kaiser_bessel_derived(2048)
fb = zafx.melfilterbank(44100, 2048, 128)
ck = zafx.cqtkernel(44100, 24, 55, 3520)
x = np.random.default_rng(0).standard_normal((8, n))


def timed(pl, a, b, B, m, reps=5):
    pl.execute(a, b, B, m)
    pl.sync()
    pl.timer_start()
    for _ in range(reps):
        pl.execute(a, b, B, m)
    return pl.timer_stop() / reps


for name, B, fwd, inv in (
        ("stft", 128, zafx.stft_plan(ham, 1024, f64=True), zafx.istft_plan(ham, 1024, f64=True)),
        ("mdct", 128, zafx.mdct_plan(kbd, f64=True), zafx.mdct_plan(kbd, inverse=True, f64=True)),
        ("mel", 128, zafx.mel_plan(ham, 1024, fb, f64=True), None),
        ("mfcc", 128, zafx.mel_plan(ham, 1024, fb, 20, f64=True), None),
        ("cqt", 32, zafx.cqt_plan(44100, 25, ck, f64=True), None)):
    d_x = zafx.DeviceBuffer.from_host(np.tile(x, (B // 8, 1)))
    T = fwd.out_dims(n)[1]
    d_s = zafx.DeviceBuffer(fwd.out_shape(B, n), fwd.out_dtype)
    ms = timed(fwd, d_x, d_s, B, n)
    txt = f"{name} f64: {B} clips {ms:.3f} ms = {B * n / ms / 1e6:.1f} Gsamples/s ({fwd.last_kernel})"
    if inv is not None:
        d_y = zafx.DeviceBuffer(inv.out_shape(B, T), inv.out_dtype)
        ims = timed(inv, d_s, d_y, B, T)
        txt += f" | inverse {ims:.3f} ms = {B * n / ims / 1e6:.1f} Gsamples/s ({inv.last_kernel})"
        d_y.free()
    print(txt, flush=True)
    d_s.free()
    d_x.free()
