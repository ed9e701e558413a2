#!/usr/bin/env python3
"""Device-resident rates of the four kinds of one window (default W = 4096: T = 217 for 10 s clips at hop W / 2) on compact and on row-padded device arrays.
usage: w4096_rates.py [clips [W]]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "zaf-python_amd"))
import zafx  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
W = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
N, H = 441000, W // 2
x = np.random.default_rng(0).standard_normal((8, N)).astype(np.float32)
d_x = zafx.DeviceBuffer.from_host(np.tile(x, (B // 8, 1)))
ham, kbd = zafx.hamming(W), zafx.kaiser_bessel_derived(W)


def rate(plan, d_in, d_out, n_in, gb, tag):
    for _ in range(30):
        plan.execute(d_in, d_out, B, n_in)
    plan.sync()
    plan.timer_start()
    for _ in range(20):
        plan.execute(d_in, d_out, B, n_in)
    ms = plan.timer_stop() / 20
    print(f"{tag:28s} {ms:7.3f} ms = {gb / ms / 8:.3f} of HBM ({plan.last_kernel})", flush=True)


for align_c, align_f, name in ((0, 0, "compact"), (16, 32, "padded")):
    fwd = zafx.stft_plan(ham, H, row_align=align_c)
    F, T = fwd.out_dims(N)
    d_s = zafx.DeviceBuffer(fwd.out_shape(B, N), np.complex64)
    rate(fwd, d_x, d_s, N, B * (4 * N + 8 * W * T) / 1e9, f"stft{W} T={T} {name}")
    inv = zafx.istft_plan(ham, H, row_align=align_c)
    d_y = zafx.DeviceBuffer((B, inv.out_dims(T)[0]), np.float32)
    rate(inv, d_s, d_y, T, B * (8 * W * T + 4 * inv.out_dims(T)[0]) / 1e9, f"istft{W} T={T} {name}")
    d_s.free(); d_y.free()
    fm = zafx.mdct_plan(kbd, row_align=align_f)
    Fm, Tm = fm.out_dims(N)
    d_m = zafx.DeviceBuffer(fm.out_shape(B, N), np.float32)
    rate(fm, d_x, d_m, N, B * (4 * N + 4 * Fm * Tm) / 1e9, f"mdct{W} T={Tm} {name}")
    im = zafx.mdct_plan(kbd, inverse=True, row_align=align_f)
    d_y = zafx.DeviceBuffer((B, im.out_dims(Tm)[0]), np.float32)
    rate(im, d_m, d_y, Tm, B * (4 * Fm * Tm + 4 * im.out_dims(Tm)[0]) / 1e9, f"imdct{W} T={Tm} {name}")
    d_m.free(); d_y.free()
