#!/usr/bin/env python3
"""W = 128 ... 2048 at frame counts that are multiples of 32 (rows on the line grid): what the smaller windows cost by themselves, without the
off-grid effects of tools/size_sweep.py (256 clips x ~10 s, hop W / 2)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "zaf-python_amd"))
import zafx
B = 256
def run(plan, d_in, n_in, reps=20):
    d_out = zafx.DeviceBuffer(plan.out_shape(B, n_in), plan.out_dtype)
    for _ in range(5): plan.execute(d_in, d_out, B, n_in)
    plan.sync(); plan.timer_start()
    for _ in range(reps): plan.execute(d_in, d_out, B, n_in)
    ms = plan.timer_stop() / reps
    return ms, d_out
for wl in (128, 256, 512, 1024, 2048):
    hop = wl // 2
    T = (441000 // hop) // 32 * 32            # frames: a multiple of 32
    n = (T - 1) * hop                          # stft: T = ceil(n / hop) + 1 ... adjusted below
    ham, kbd = zafx.hamming(wl), zafx.kaiser_bessel_derived(wl)
    x = np.random.default_rng(0).standard_normal((8, n)).astype(np.float32)
    d_x = zafx.DeviceBuffer.from_host(np.tile(x, (B // 8, 1)))
    p = zafx.stft_plan(ham, hop); Ts = p.out_dims(n)[1]
    ms, d_s = run(p, d_x, n); by = B * n * 4 + d_s.nbytes
    pi = zafx.istft_plan(ham, hop); ms2, d_y = run(pi, d_s, Ts); by2 = d_s.nbytes + d_y.nbytes
    pm = zafx.mdct_plan(kbd); Tm = pm.out_dims(n)[1]
    ms3, d_m = run(pm, d_x, n); by3 = B * n * 4 + d_m.nbytes
    pim = zafx.mdct_plan(kbd, inverse=True); ms4, d_z = run(pim, d_m, Tm); by4 = d_m.nbytes + d_z.nbytes
    print(f"W={wl:5d} n={n} stft T={Ts} {ms:.3f} ms {by/ms/1e9:.2f} TB/s ({p.last_kernel}) | istft {ms2:.3f} {by2/ms2/1e9:.2f} | mdct T={Tm} {ms3:.3f} {by3/ms3/1e9:.2f} ({pm.last_kernel}) | imdct {ms4:.3f} {by4/ms4/1e9:.2f}", flush=True)
    for b in (d_x, d_s, d_y, d_m, d_z): b.free()
