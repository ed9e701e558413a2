#!/usr/bin/env python3
"""Per-phase cycle counters of the persistent kernels (library built with -DZAFX_PROF).

    tools/build_prof.sh && ZAFX_LIBRARY=tools/bin/libzafx_prof.so python tools/prof_phases.py istft|mdct|imdct
"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "zaf-python_amd"))
import zafx  # noqa: E402
from zafx import _lib  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "istft"
lib = _lib.load()
B, N, W, H = 1024, 441000, 2048, 1024
x = np.random.default_rng(0).standard_normal((8, N)).astype(np.float32)
d_x = zafx.DeviceBuffer.from_host(np.tile(x, (B // 8, 1)))
if kind == "istft":
    w = zafx.hamming(W)
    fwd, plan = zafx.stft_plan(w, H), zafx.istft_plan(w, H)
    F, T = fwd.out_dims(N)
    d_in = zafx.DeviceBuffer((B, F, T), np.complex64)
    fwd.execute(d_x, d_in, B, N)
    fwd.sync()
    n_in, tiles = T, 27
    d_out = zafx.DeviceBuffer((B, plan.out_dims(T)[0]), np.float32)
elif kind == "istft8192":
    w = zafx.hamming(8192)
    N = 4096 * 111
    x = np.random.default_rng(0).standard_normal((8, N)).astype(np.float32)
    d_x.free()
    d_x = zafx.DeviceBuffer.from_host(np.tile(x, (B // 8, 1)))
    fwd, plan = zafx.stft_plan(w, 4096), zafx.istft_plan(w, 4096)
    F, T = fwd.out_dims(N)
    d_in = zafx.DeviceBuffer((B, F, T), np.complex64)
    fwd.execute(d_x, d_in, B, N)
    fwd.sync()
    n_in, tiles = T, 14
    d_out = zafx.DeviceBuffer((B, plan.out_dims(T)[0]), np.float32)
elif kind == "imdct8192":
    w = zafx.kaiser_bessel_derived(8192)
    N = 4096 * 111
    x = np.random.default_rng(0).standard_normal((8, N)).astype(np.float32)
    d_x.free()
    d_x = zafx.DeviceBuffer.from_host(np.tile(x, (B // 8, 1)))
    fwd, plan = zafx.mdct_plan(w), zafx.mdct_plan(w, inverse=True)
    F, T = fwd.out_dims(N)
    d_in = zafx.DeviceBuffer((B, F, T), np.float32)
    fwd.execute(d_x, d_in, B, N)
    fwd.sync()
    n_in, tiles = T, (T + 15) // 16
    d_out = zafx.DeviceBuffer((B, plan.out_dims(T)[0]), np.float32)
elif kind in ("stft", "stft1"):
    plan = zafx.stft_plan(zafx.hamming(W), H, onesided=kind == "stft1")
    F, T = plan.out_dims(N)
    d_in, n_in, tiles = d_x, N, 27
    d_out = zafx.DeviceBuffer((B, F, T), np.complex64)
elif kind == "mdct":
    plan = zafx.mdct_plan(zafx.kaiser_bessel_derived(W))
    F, T = plan.out_dims(N)
    d_in, n_in, tiles = d_x, N, (T + 31) // 32
    d_out = zafx.DeviceBuffer((B, F, T), np.float32)
elif kind == "imdct":
    w = zafx.kaiser_bessel_derived(W)
    fwd, plan = zafx.mdct_plan(w), zafx.mdct_plan(w, inverse=True)
    F, T = fwd.out_dims(N)
    d_in = zafx.DeviceBuffer((B, F, T), np.float32)
    fwd.execute(d_x, d_in, B, N)
    fwd.sync()
    n_in, tiles = T, (T + 31) // 32
    d_out = zafx.DeviceBuffer((B, plan.out_dims(T)[0]), np.float32)
elif kind == "spec":   # the |X| rows of the STFT on k_mel2 (MODE 2)
    plan = zafx.stft_plan(zafx.hamming(W), H, onesided="magnitude")
    F, T = plan.out_dims(N)
    d_in, n_in, tiles = d_x, N, 27
    d_out = zafx.DeviceBuffer((B, F, T), np.float32)
elif kind in ("mel", "mfcc"):
    w = zafx.hamming(W)
    fb = zafx.melfilterbank(44100, W, 128)
    plan = zafx.mel_plan(w, H, fb, 20 if kind == "mfcc" else None)
    F, T = plan.out_dims(N)
    d_in, n_in, tiles = d_x, N, 27
    d_out = zafx.DeviceBuffer((B, F, T), np.float32)
elif kind in ("mel64", "mfcc64"):
    w = zafx.hamming(W)
    fb = zafx.melfilterbank(44100, W, 128)
    plan = zafx.mel_plan(w, H, fb, 20 if kind == "mfcc64" else None, f64=True)
    B = 256
    d_x.free()
    d_x = zafx.DeviceBuffer.from_host(np.tile(x.astype(np.float64), (B // 8, 1)))
    F, T = plan.out_dims(N)
    d_in, n_in, tiles = d_x, N, 27
    d_out = zafx.DeviceBuffer((B, F, T), np.float64)
elif kind == "cqt64":
    B, N = 64, 1323000
    x = np.random.default_rng(0).standard_normal((8, N))
    d_x.free()
    d_in = zafx.DeviceBuffer.from_host(np.tile(x, (B // 8, 1)))
    plan = zafx.cqt_plan(44100, 25, zafx.cqtkernel(44100, 24, 55, 3520), f64=True)
    F, T = plan.out_dims(N)
    n_in, tiles = N, T * 1.0
    d_out = zafx.DeviceBuffer((B, F, T), np.float64)
elif kind == "cqt":
    B, N = 128, 1323000
    x = np.random.default_rng(0).standard_normal((8, N)).astype(np.float32)
    d_in = zafx.DeviceBuffer.from_host(np.tile(x, (B // 8, 1)))
    plan = zafx.cqt_plan(44100, 25, zafx.cqtkernel(44100, 24, 55, 3520))
    F, T = plan.out_dims(N)
    n_in, tiles = N, T * 1.0   # marks are per frame here
    d_out = zafx.DeviceBuffer((B, F, T), np.float32)
else:
    raise SystemExit("kind must be istft, mdct, imdct or cqt")
name = "zafx_debug_prof_" + {"mfcc": "mel", "stft1": "stft", "spec": "mel", "mfcc64": "mel64", "istft8192": "istft", "imdct8192": "imdct"}.get(kind, kind)
fn = getattr(lib, name)
out = (ctypes.c_ulonglong * 16)()
# optional second argument: the waves to time, e.g. "0,5,15" or "all" (default: wave 1)
waves = sys.argv[2] if len(sys.argv) > 2 else "1"
waves = range(16) if waves == "all" else [int(v) for v in waves.split(",")]
for wv in waves:
    getattr(lib, name + "_thread")(wv * 64)
    plan.execute(d_in, d_out, B, n_in)
    plan.sync()
    fn(out)
    reps = 5
    for _ in range(reps):
        plan.execute(d_in, d_out, B, n_in)
    plan.sync()
    fn(out)
    per_tile = reps * (tiles * B / 256 if kind == 'cqt64' else 16 if kind == 'cqt' else tiles * B / 256)   # cqt: workgroup 7 = 16 frames per launch, counts are per frame
    print(kind, f"wave {wv}: cycles per tile between marks:", " | ".join(f"{i}:{out[i] / per_tile:.0f}" for i in range(12)),
          "| total", round(sum(out) / per_tile))
