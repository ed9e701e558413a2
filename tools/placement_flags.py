#!/usr/bin/env python3
"""The STFT of config 2 into outputs allocated with hipExtMallocWithFlags (default / fine-grained / uncached / contiguous), several
of each held at once: does any allocation flavour avoid the slow placements of DESIGN.md 3?   python tools/placement_flags.py [n]"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zaf-python_amd"))
import zafx  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")
B, N, W, H = 1024, 441000, 2048, 1024
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
x = np.stack([np.random.default_rng([0, c]).standard_normal(N).astype(np.float32) for c in range(8)])
d_in = zafx.DeviceBuffer.from_host(np.tile(x, (B // 8, 1)))
plan = zafx.stft_plan(zafx.hamming(W), H)
shape = plan.out_shape(B, N)
nbytes = int(np.prod(shape)) * 8


def probe(buf, reps=10):
    for _ in range(3):
        plan.execute(d_in, buf, B, N)
    plan.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        plan.execute(d_in, buf, B, N)
    plan.sync()
    return (time.perf_counter() - t0) / reps * 1e3


warm = zafx.DeviceBuffer(shape, np.complex64)
t_end = time.perf_counter() + 0.5
while time.perf_counter() < t_end:
    probe(warm, 4)
held = [warm]
print("hipMalloc (first):", f"{probe(warm):.4f}")
def alloc(flag, shp=shape, dtype=np.complex64):
    p = ctypes.c_void_p()
    nb = int(np.prod(shp)) * np.dtype(dtype).itemsize
    rc = hip.hipExtMallocWithFlags(ctypes.byref(p), ctypes.c_size_t(nb), ctypes.c_uint(flag))
    if rc != 0:
        raise RuntimeError(f"hipExtMallocWithFlags rc={rc}")
    return zafx.DeviceBuffer(shp, dtype, 0, _ptr_from_pool=p)


mode = sys.argv[2] if len(sys.argv) > 2 else "flavours"
if mode == "flavours":
    for name, flag in (("default", 0), ("contiguous", 4), ("finegrained", 1), ("uncached", 3), ("default again", 0)):
        times = []
        for _ in range(n):
            buf = alloc(flag)
            held.append(buf)
            times.append(f"{probe(buf):.4f}")
        print(f"{name:14s}", " ".join(times), flush=True)
else:
    # default and fine-grained allocations interleaved, all held: is fine-grained memory fast wherever it lands?
    rows = {0: [], 1: []}
    for i in range(2 * n):
        buf = alloc(i & 1)
        held.append(buf)
        rows[i & 1].append(probe(buf))
    print("default    ", " ".join(f"{t:.4f}" for t in rows[0]))
    print("finegrained", " ".join(f"{t:.4f}" for t in rows[1]), flush=True)
    # the other kernels with a fine-grained 2-D side: mdct output, istft / imdct input
    kbd = zafx.kaiser_bessel_derived(W)

    def timed(pl, a, b, m, reps=20):
        for _ in range(30):
            pl.execute(a, b, B, m)
        pl.sync()
        pl.timer_start()
        for _ in range(reps):
            pl.execute(a, b, B, m)
        return pl.timer_stop() / reps

    for b in held[1:]:
        b.free()
    held[:] = held[:1]
    T = plan.out_dims(N)[1]
    for flag in (0, 1, 0, 1):
        spec = alloc(flag)
        t_stft = timed(plan, d_in, spec, N)
        inv = zafx.istft_plan(zafx.hamming(W), H)
        y = alloc(flag, inv.out_shape(B, T), np.float32)
        t_istft = timed(inv, spec, y, T)
        md = zafx.mdct_plan(kbd)
        coef = alloc(flag, md.out_shape(B, N), np.float32)
        t_mdct = timed(md, d_in, coef, N)
        imd = zafx.mdct_plan(kbd, inverse=True)
        y2 = alloc(flag, imd.out_shape(B, T), np.float32)
        t_imdct = timed(imd, coef, y2, T)
        fb = zafx.melfilterbank(44100, W, 128)
        mel = zafx.mel_plan(zafx.hamming(W), H, fb)
        m_out = alloc(flag, mel.out_shape(B, N), np.float32)
        t_mel = timed(mel, d_in, m_out, N)
        print(f"flag {flag}: stft {t_stft:.4f} istft {t_istft:.4f} mdct {t_mdct:.4f} imdct {t_imdct:.4f} mel {t_mel:.4f}", flush=True)
        held.extend([spec, y, coef, y2, m_out])
