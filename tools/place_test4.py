#!/usr/bin/env python3
"""Does the placement of the INPUT matter for the kernels that read it linearly (mel, mdct) -- eight allocations, each timed.
   gpurun -- 'python tools/place_test4.py'"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "zaf-python_amd"))
import zafx
B, N, W, H = 1024, 441000, 2048, 1024
x = np.random.default_rng(0).standard_normal((8, N)).astype(np.float32)
host = np.tile(x, (B // 8, 1))
ins = [zafx.DeviceBuffer((B, N), np.float32) for _ in range(8)]
for d in ins:
    d.upload(host)
def t(plan, d_x, d_o, reps=30):
    for _ in range(6): plan.execute(d_x, d_o, B, N)
    plan.sync()
    t0 = time.perf_counter()
    for _ in range(reps): plan.execute(d_x, d_o, B, N)
    plan.sync()
    return (time.perf_counter() - t0) / reps * 1e3
mel = zafx.mel_plan(zafx.hamming(W), H, zafx.melfilterbank(44100, W, 128))
d_m = zafx.DeviceBuffer((B,) + tuple(mel.out_dims(N)), np.float32)
mdct = zafx.mdct_plan(zafx.kaiser_bessel_derived(W))
outs = [zafx.DeviceBuffer((B,) + tuple(mdct.out_dims(N)), np.float32) for _ in range(8)]
t(mel, ins[0], d_m); t(mel, ins[0], d_m)
print("mel  by input :", " ".join(f"{t(mel, d, d_m):.4f}" for d in ins))
print("mdct by input :", " ".join(f"{t(mdct, d, outs[0]):.4f}" for d in ins))
print("mdct by output:", " ".join(f"{t(mdct, ins[0], o):.4f}" for o in outs))
