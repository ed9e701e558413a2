#!/usr/bin/env python3
"""Where the 7.25 GB spectrum lands decides between 1.50 and 1.70 ms for config 2: what decides it?

    python tools/placement.py [n_candidates] [variant ...]

Allocates `n_candidates` output buffers of config 2 (held at once), and times into each of them
  * the shipped STFT (reference layout: 128-byte runs at a 3456-byte row stride),
  * the same STFT from every variant library named (tools/bin/libzafx_<variant>.so, tools/build_variant.sh),
  * the frame-major STFT (16 KB contiguous runs) and a plain hipMemset (linear writes) -- does the mode belong to the
    buffer for EVERY write pattern, or only to the row-strided one?
and prints what the driver exposes about memory (partition modes, page-table fragment size, VRAM manager state when debugfs
is there).  Replaces round 2's place_test*.py scratch scripts; results are quoted in profiles/r03_notes.md.
"""
import ctypes
import glob
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zaf-python_amd"))
import zafx  # noqa: E402
from zafx import _lib  # noqa: E402

B, N, W, H = 1024, 441000, 2048, 1024


def show(path):
    for p in sorted(glob.glob(path)):
        try:
            with open(p) as f:
                txt = f.read().strip()
            print(f"  {p}: {txt[:400]}")
        except OSError as exc:
            print(f"  {p}: <{exc.__class__.__name__}>")


def driver_facts():
    print("driver / topology:")
    show("/sys/class/drm/card*/device/current_memory_partition")
    show("/sys/class/drm/card*/device/current_compute_partition")
    show("/sys/class/drm/card*/device/mem_info_vram_total")
    show("/sys/class/drm/card*/device/mem_info_vram_used")
    show("/sys/module/amdgpu/parameters/vm_fragment_size")
    show("/sys/module/amdgpu/parameters/vm_block_size")
    show("/sys/module/amdgpu/parameters/vm_size")
    show("/sys/module/amdgpu/parameters/mtype_local")
    show("/sys/class/kfd/kfd/topology/nodes/*/mem_banks/*/properties")
    if not os.path.isdir("/sys/kernel/debug/dri"):
        os.system("mount -t debugfs none /sys/kernel/debug 2>/dev/null")
    for p in sorted(glob.glob("/sys/kernel/debug/dri/*/amdgpu_vram_mm"))[:1]:
        print("  --", p)
        os.system(f"head -60 {p}")


class RawPlan:
    """An STFT plan of ANOTHER build of the library (variant .so), driven through raw ctypes; device pointers are shared."""

    def __init__(self, path, layout=0):
        self.lib = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)
        for name, (res, args) in _lib.SYMBOLS.items():
            if hasattr(self.lib, name):
                fn = getattr(self.lib, name)
                fn.restype, fn.argtypes = res, args
        prm = _lib.ZafxParams()
        prm.struct_size = ctypes.sizeof(prm)
        prm.window_length, prm.step_length, prm.layout = W, H, layout
        self.h = ctypes.c_void_p()
        assert self.lib.zafx_plan_create(ctypes.byref(self.h), 0, _lib.STFT, ctypes.byref(prm)) == 0, self.lib.zafx_last_error()
        w = np.ascontiguousarray(zafx.hamming(W), dtype=np.float32)
        assert self.lib.zafx_plan_set_constant(self.h, _lib.CONST_WINDOW, ctypes.c_void_p(w.ctypes.data), w.nbytes) == 0

    def execute(self, d_in, d_out, b, n):
        assert self.lib.zafx_execute(self.h, d_in.ptr, d_out.ptr, b, n) == 0, self.lib.zafx_last_error()

    def sync(self):
        assert self.lib.zafx_sync(self.h) == 0


def probe(plan, d_in, d_out, reps=10):
    for _ in range(3):
        plan.execute(d_in, d_out, B, N)
    plan.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        plan.execute(d_in, d_out, B, N)
    plan.sync()
    return (time.perf_counter() - t0) / reps * 1e3


_SYNC = []


def memset_gbs(buf, reps=3):
    if not _SYNC:
        _SYNC.append(zafx.DeviceBuffer((4,), np.uint8))
    buf.fill_zero()
    _SYNC[0].download()              # (a blocking copy on the null stream: the memsets before it are done)
    t0 = time.perf_counter()
    for _ in range(reps):
        buf.fill_zero()
    _SYNC[0].download()
    return buf.nbytes * reps / (time.perf_counter() - t0) / 1e9


def main():
    n_cand = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    variants = sys.argv[2:]
    driver_facts()
    x = np.stack([np.random.default_rng([0, c]).standard_normal(N).astype(np.float32) for c in range(8)])
    d_in = zafx.DeviceBuffer.from_host(np.tile(x, (B // 8, 1)))
    plans = {"ft": zafx.stft_plan(zafx.hamming(W), H), "tf": zafx.stft_plan(zafx.hamming(W), H, layout="TF")}
    for v in variants:
        plans[v] = RawPlan(os.path.join(ROOT, "tools", "bin", f"libzafx_{v}.so"))
    shape = plans["ft"].out_shape(B, N)
    # clocks up before anything is compared
    warm = zafx.DeviceBuffer(shape, np.complex64)
    t_end = time.perf_counter() + 0.5
    while time.perf_counter() < t_end:
        probe(plans["ft"], d_in, warm, reps=4)
    bufs = [warm]
    for _ in range(n_cand - 1):
        try:
            bufs.append(zafx.DeviceBuffer(shape, np.complex64))
        except zafx.ZafxError as exc:
            print("allocation stopped:", exc)
            break
    names = list(plans)
    print(f"{'#':>3} {'address':>16} " + " ".join(f"{n:>9}" for n in names) + "  memset GB/s")
    rows = []
    for i, b in enumerate(bufs):
        t = [probe(plans[n], d_in, b) for n in names]
        ms = memset_gbs(b)
        rows.append(t)
        print(f"{i:3d} {b.ptr.value:#16x} " + " ".join(f"{v:9.4f}" for v in t) + f"  {ms:8.0f}")
    a = np.asarray(rows)
    print("corr with ft:", " ".join(f"{n}={np.corrcoef(a[:, 0], a[:, j])[0, 1]:+.2f}" for j, n in enumerate(names) if j))
    print("min / max per plan:", " ".join(f"{n}={a[:, j].min():.4f}/{a[:, j].max():.4f}" for j, n in enumerate(names)))
    # second pass over the first six, in reverse order: is a buffer's mode stable in time?
    print("again (reverse):", " ".join(f"{probe(plans['ft'], d_in, b):.4f}" for b in reversed(bufs[:6])))


if __name__ == "__main__":
    main()
