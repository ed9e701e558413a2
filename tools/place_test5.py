#!/usr/bin/env python3
"""One 72 GiB allocation, the STFT's output placed every 8 GiB inside it: do the placement modes follow allocations or regions?
   gpurun -- 'python tools/place_test5.py'"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "zaf-python_amd"))
import zafx
B, N, W, H = 1024, 441000, 2048, 1024
x = np.random.default_rng(0).standard_normal((8, N)).astype(np.float32)
plan = zafx.stft_plan(zafx.hamming(W), H)
F, T = plan.out_dims(N)
d_x = zafx.DeviceBuffer.from_host(np.tile(x, (B // 8, 1)))
arena = zafx.DeviceBuffer((72 << 30,), np.uint8)
base = arena.ptr.value
def t(off, reps=12):
    d_o = zafx.DeviceBuffer((B, F, T), np.complex64, _ptr_from_pool=ctypes.c_void_p(base + off))
    for _ in range(4): plan.execute(d_x, d_o, B, N)
    plan.sync()
    t0 = time.perf_counter()
    for _ in range(reps): plan.execute(d_x, d_o, B, N)
    plan.sync()
    d_o.ptr = ctypes.c_void_p()
    return (time.perf_counter() - t0) / reps * 1e3
t(0); t(0)
print("output every 8 GiB inside one 72 GiB allocation:", " ".join(f"{t(k << 33):.3f}" for k in range(8)))
print("every 1 GiB from 0:", " ".join(f"{t(k << 30):.3f}" for k in range(16)))
