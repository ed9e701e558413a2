#!/usr/bin/env python3
"""Does a buffer built from SMALL, SCATTERED physical chunks have a write mode of its own?  (tools/exp_vmmalloc.hip)

    python tools/placement_vmm.py [n_hipmalloc]

Times config 2's STFT into (a) n hipMalloc allocations held at once, (b) buffers assembled through the virtual-memory API from
chunks of 2 MB ... 1 GB mapped in creation order, reversed, interleaved or shuffled.  Results: profiles/r03_notes.md 11.
"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zaf-python_amd"))
import zafx  # noqa: E402

B, N, W, H = 1024, 441000, 2048, 1024


def probe(plan, d_in, d_out, reps=10):
    for _ in range(3):
        plan.execute(d_in, d_out, B, N)
    plan.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        plan.execute(d_in, d_out, B, N)
    plan.sync()
    return (time.perf_counter() - t0) / reps * 1e3


class Foreign(zafx.DeviceBuffer):
    """A DeviceBuffer view of memory this script owns (never handed to zafx_free)."""

    def free(self):
        self.ptr = ctypes.c_void_p()


def main():
    n_malloc = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    vmm = ctypes.CDLL(os.path.join(ROOT, "tools", "bin", "libvmmalloc.so"))
    vmm.vmm_alloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint, ctypes.c_int]
    vmm.vmm_free.argtypes = [ctypes.c_void_p]
    vmm.vmm_granularity.restype = ctypes.c_size_t
    x = np.stack([np.random.default_rng([0, c]).standard_normal(N).astype(np.float32) for c in range(8)])
    d_in = zafx.DeviceBuffer.from_host(np.tile(x, (B // 8, 1)))
    plan = zafx.stft_plan(zafx.hamming(W), H)
    shape = plan.out_shape(B, N)
    nbytes = int(np.prod(shape)) * 8
    print(f"granularity min {vmm.vmm_granularity(0)} recommended {vmm.vmm_granularity(1)}; buffer {nbytes / 2**30:.2f} GiB")
    warm = zafx.DeviceBuffer(shape, np.complex64)
    t_end = time.perf_counter() + 0.5
    while time.perf_counter() < t_end:
        probe(plan, d_in, warm, reps=4)
    bufs = [warm] + [zafx.DeviceBuffer(shape, np.complex64) for _ in range(n_malloc - 1)]
    print("hipMalloc:", " ".join(f"{probe(plan, d_in, b):.4f}" for b in bufs))
    names = {0: "in order", 1: "shuffled", 2: "reversed", 3: "interleaved"}
    cases = [(1 << 30, 0, 100), (1 << 30, 1, 100), (128 << 20, 0, 100), (128 << 20, 1, 100), (16 << 20, 0, 100), (16 << 20, 1, 100),
             (16 << 20, 3, 100), (2 << 20, 0, 100), (2 << 20, 1, 100), (2 << 20, 2, 100), (2 << 20, 1, 300), (16 << 20, 1, 300)]
    for rnd in range(2):
        for chunk, order, pool in cases:
            p = ctypes.c_void_p()
            t0 = time.perf_counter()
            rc = vmm.vmm_alloc(ctypes.byref(p), nbytes, chunk, order, 1234 + rnd, pool)
            t_alloc = time.perf_counter() - t0
            if rc:
                print(f"chunk {chunk >> 20} MB {names[order]}: allocation failed")
                continue
            b = Foreign(shape, np.complex64, _ptr_from_pool=p)
            t = [probe(plan, d_in, b) for _ in range(2)]
            t0 = time.perf_counter()
            vmm.vmm_free(p)
            print(f"round {rnd} chunk {chunk >> 20:5d} MB {names[order]:11s} pool {pool:3d}%: {t[0]:.4f} {t[1]:.4f} ms"
                  f"   (alloc {t_alloc * 1e3:.0f} ms, free {(time.perf_counter() - t0) * 1e3:.0f} ms)", flush=True)
    print("hipMalloc again:", " ".join(f"{probe(plan, d_in, b):.4f}" for b in bufs))
    # with the hipMalloc candidates released: the chunks now come from the memory those occupied
    for b in bufs[1:]:
        b.free()
    for chunk, order, pool in [(2 << 20, 1, 100), (16 << 20, 1, 100), (1 << 30, 0, 100)]:
        p = ctypes.c_void_p()
        if vmm.vmm_alloc(ctypes.byref(p), nbytes, chunk, order, 99, pool):
            continue
        b = Foreign(shape, np.complex64, _ptr_from_pool=p)
        print(f"after frees: chunk {chunk >> 20:5d} MB {names[order]:11s}: {probe(plan, d_in, b):.4f} {probe(plan, d_in, b):.4f} ms", flush=True)
        vmm.vmm_free(p)


if __name__ == "__main__":
    main()
