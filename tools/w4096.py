#!/usr/bin/env python3
"""W = 4096 / 8192 in the reference layout, device resident: rate of the shipped kernels (k_stft_ft16b: two bands of bins
per 16-frame tile) and, when tools/bin/libzafx_<variant>.so exist (tools/build_variant.sh), of those builds.

    python tools/w4096.py [variant ...]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zaf-python_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import zafx  # noqa: E402

B = 1024


def rate(plan, d_in, d_out, n, reps=20):
    for _ in range(5):
        plan.execute(d_in, d_out, B, n)
    plan.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        plan.execute(d_in, d_out, B, n)
    plan.sync()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    rng = np.random.default_rng(5)
    for n in (441000, 441000 + 2048, 2048 * 223):
        x = rng.standard_normal((8, n)).astype(np.float32)
        d_in = zafx.DeviceBuffer.from_host(np.tile(x, (B // 8, 1)))
        for wl, hop in ((2048, 1024), (4096, 2048), (4096, 1024), (8192, 4096)):
            for kw in ({}, {"onesided": True}, {"onesided": "magnitude"}):
                plan = zafx.stft_plan(zafx.hamming(wl), hop, **kw)
                shape = plan.out_shape(B, n)
                d_out = zafx.DeviceBuffer(shape, plan.out_dtype)
                t_end = time.perf_counter() + 0.3
                while time.perf_counter() < t_end:
                    rate(plan, d_in, d_out, n, reps=4)
                ms = rate(plan, d_in, d_out, n)
                nbytes = B * n * 4 + d_out.nbytes
                print(f"n {n} W {wl} hop {hop} {str(kw):28s} {plan.last_kernel:14s} T {shape[-1]:4d}: {ms:7.3f} ms  {nbytes / ms / 1e9:6.2f} TB/s", flush=True)
                d_out.free()
        for wl, hop in ((2048, 1024), (4096, 2048), (4096, 1024), (8192, 4096)):
            fb = zafx.melfilterbank(44100, wl, 128)
            for ncoef in (None, 20):
                plan = zafx.mel_plan(zafx.hamming(wl), hop, fb, ncoef)
                d_out = zafx.DeviceBuffer(plan.out_shape(B, n), plan.out_dtype)
                rate(plan, d_in, d_out, n, reps=20)
                ms = rate(plan, d_in, d_out, n)
                print(f"n {n} W {wl} hop {hop} {'mfcc' if ncoef else 'mel ':28s} {plan.last_kernel:14s}: {ms:7.3f} ms  {B * n / ms / 1e6:7.1f} Gsamples/s", flush=True)
                d_out.free()
        for wl in (2048, 4096, 8192):
            kbd = zafx.kaiser_bessel_derived(wl)
            fwd, inv = zafx.mdct_plan(kbd), zafx.mdct_plan(kbd, inverse=True)
            d_c = zafx.DeviceBuffer(fwd.out_shape(B, n), fwd.out_dtype)
            T = fwd.out_dims(n)[1]
            d_y = zafx.DeviceBuffer(inv.out_shape(B, T), inv.out_dtype)
            rate(fwd, d_in, d_c, n, reps=20)
            ms = rate(fwd, d_in, d_c, n)
            print(f"n {n} W {wl} mdct  {fwd.last_kernel:14s} T {T}: {ms:7.3f} ms  {(B * n * 4 + d_c.nbytes) / ms / 1e9:6.2f} TB/s", flush=True)
            rate(inv, d_c, d_y, T, reps=20)
            ms = rate(inv, d_c, d_y, T)
            print(f"n {n} W {wl} imdct {inv.last_kernel:14s} T {T}: {ms:7.3f} ms  {(d_y.nbytes + d_c.nbytes) / ms / 1e9:6.2f} TB/s", flush=True)
            d_c.free(); d_y.free()
        for wl, hop in ((2048, 1024), (4096, 2048), (4096, 1024), (8192, 4096)):
            fwd, inv = zafx.stft_plan(zafx.hamming(wl), hop), zafx.istft_plan(zafx.hamming(wl), hop)
            d_c = zafx.DeviceBuffer(fwd.out_shape(B, n), fwd.out_dtype)
            T = fwd.out_dims(n)[1]
            d_y = zafx.DeviceBuffer(inv.out_shape(B, T), inv.out_dtype)
            fwd.execute(d_in, d_c, B, n)
            rate(inv, d_c, d_y, T, reps=20)
            ms = rate(inv, d_c, d_y, T)
            print(f"n {n} W {wl} hop {hop} istft {inv.last_kernel:14s} T {T}: {ms:7.3f} ms  {(d_y.nbytes + d_c.nbytes) / ms / 1e9:6.2f} TB/s", flush=True)
            d_c.free(); d_y.free()
        d_in.free()


if __name__ == "__main__":
    main()
