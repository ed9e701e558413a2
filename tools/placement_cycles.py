#!/usr/bin/env python3
"""Do the slow placements of DESIGN.md 3 survive a free / re-allocate cycle?  n outputs of config 2 are allocated (all held), probed
with the STFT, freed; repeated.   python tools/placement_cycles.py [n] [cycles]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zaf-python_amd"))
import zafx  # noqa: E402

B, N, W, H = 1024, 441000, 2048, 1024
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 4
x = np.stack([np.random.default_rng([0, c]).standard_normal(N).astype(np.float32) for c in range(8)])
d_in = zafx.DeviceBuffer.from_host(np.tile(x, (B // 8, 1)))
plan = zafx.stft_plan(zafx.hamming(W), H)
shape = plan.out_shape(B, N)


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
print("first allocation:", f"{probe(warm):.4f}", hex(warm.ptr.value))
for c in range(cycles):
    bufs = [zafx.DeviceBuffer(shape, np.complex64) for _ in range(n)]
    print(f"cycle {c}:", " ".join(f"{probe(b):.3f}" for b in bufs), "| first again", f"{probe(warm):.3f}", flush=True)
    for b in bufs:
        b.free()
    time.sleep(float(sys.argv[3]) if len(sys.argv) > 3 else 1.0)
