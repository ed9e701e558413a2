#!/usr/bin/env python3
"""Counter comparison of a slow and a fast placement of config 2's spectrum (run under `rocprofv3 --pmc ... --kernel-trace`):
allocates `n` outputs, probes each, then launches the STFT 6 times into the slowest and 6 times into the fastest.  The last
12 dispatches of k_stft_ft16 in the counter CSV are those (slow first).  See tools/placement.py, profiles/r03_notes.md."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zaf-python_amd"))
import zafx  # noqa: E402

B, N, W, H = 1024, 441000, 2048, 1024
n = int(sys.argv[1]) if len(sys.argv) > 1 else 14
x = np.stack([np.random.default_rng([0, c]).standard_normal(N).astype(np.float32) for c in range(8)])
d_in = zafx.DeviceBuffer.from_host(np.tile(x, (B // 8, 1)))
plan = zafx.stft_plan(zafx.hamming(W), H)
shape = plan.out_shape(B, N)
bufs, times = [], []
for i in range(n):
    bufs.append(zafx.DeviceBuffer(shape, np.complex64))
    for _ in range(2 if i else 30):
        plan.execute(d_in, bufs[-1], B, N)
    plan.sync()
    plan.timer_start()
    for _ in range(4):
        plan.execute(d_in, bufs[-1], B, N)
    times.append(plan.timer_stop() / 4)
slow, fast = int(np.argmax(times)), int(np.argmin(times))
print("probe ms:", " ".join(f"{t:.3f}" for t in times), f"-> slow #{slow} fast #{fast}")
for b in (bufs[slow], bufs[fast]):
    for _ in range(6):
        plan.execute(d_in, b, B, N)
    plan.sync()
print("done")
