#!/usr/bin/env python3
"""Does the placement of the buffers in the address space change the STFT's rate?  One arena, the input and the output placed at
chosen offsets inside it.   gpurun -- 'python tools/place_test.py'"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "zaf-python_amd"))
import zafx  # noqa: E402

B, N, W, H = 1024, 441000, 2048, 1024
x = np.random.default_rng(0).standard_normal((8, N)).astype(np.float32)
plan = zafx.stft_plan(zafx.hamming(W), H)
F, T = plan.out_dims(N)
in_bytes, out_bytes = B * N * 4, B * F * T * 8
arena = zafx.DeviceBuffer((in_bytes + out_bytes + (3 << 30),), np.uint8)
base = arena.ptr.value
print(f"arena at {base:#x}, input {in_bytes / 2**30:.3f} GiB, output {out_bytes / 2**30:.3f} GiB")
host = np.tile(x, (B // 8, 1))


def view(off, shape, dtype):
    return zafx.DeviceBuffer(shape, dtype, _ptr_from_pool=ctypes.c_void_p(base + off))


def run(in_off, out_off):
    d_x, d_o = view(in_off, (B, N), np.float32), view(out_off, (B, F, T), np.complex64)
    d_x.upload(host)
    for _ in range(20):
        plan.execute(d_x, d_o, B, N)
    plan.sync()
    t0 = time.perf_counter()
    for _ in range(60):
        plan.execute(d_x, d_o, B, N)
    plan.sync()
    ms = (time.perf_counter() - t0) / 60 * 1e3
    print(f"input at +{in_off:#12x}  output at +{out_off:#12x} ({(base + out_off) % (1 << 32):#011x} mod 4 GiB): {ms:.4f} ms")
    d_x.ptr = ctypes.c_void_p()   # views: nothing to free
    d_o.ptr = ctypes.c_void_p()


al = lambda v, a: (v + a - 1) // a * a
o0 = al(in_bytes, 1 << 30)
for out_off in (o0, o0 + (4 << 10), o0 + (64 << 10), o0 + (1 << 20), o0 + (2 << 20), o0 + (16 << 20), o0 + (128 << 20), o0 + (512 << 20),
                o0 + (1 << 30), o0 + (1 << 30) + (256 << 20), o0):
    run(0, out_off)
for in_off in (4 << 10, 2 << 20, 256 << 20):
    run(in_off, o0 + (1 << 30))
