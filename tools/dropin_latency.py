#!/usr/bin/env python3
"""Latency of the drop-in single-clip calls (BASELINE config 1: one 10 s clip), host arrays in and out.  (The NumPy path's times for
the same calls are `cpu_baseline` of bench.py: the oracle is not imported outside tests/, smoke() and that leg.)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "zaf-python_amd"))
import zafx  # noqa: E402

x = np.random.default_rng(0).standard_normal(441000)
w = zafx.hamming(2048)
fb = zafx.melfilterbank(44100, 2048, 128)


def bench(fn, n=20):
    fn()
    fn()
    t = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    return np.median(t) * 1e3


s = zafx.stft(x, w, 1024)
print("stft   zafx %.2f ms" % bench(lambda: zafx.stft(x, w, 1024)))
print("istft  zafx %.2f ms" % bench(lambda: zafx.istft(s, w, 1024)))
print("mel    zafx %.2f ms" % bench(lambda: zafx.melspectrogram(x, w, 1024, fb)))
print("mdct   zafx %.2f ms" % bench(lambda: zafx.mdct(x, zafx.kaiser_bessel_derived(2048))))
