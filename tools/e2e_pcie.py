#!/usr/bin/env python3
"""Host-array boundary including PCIe (Plan.run_host = zafx_run_host, the three-stage pipeline): chunk size sweep,
page-locked and pageable arrays, two-sided and one-sided STFT.   python tools/e2e_pcie.py [clips]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zaf-python_amd"))
import zafx  # noqa: E402

clips = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N, W, H = 441000, 2048, 1024
base = np.stack([np.random.default_rng([0, c]).standard_normal(N).astype(np.float32) for c in range(8)])
x = zafx.pinned_empty((clips, N), np.float32)
x[:] = np.tile(base, (clips // 8, 1))
for onesided in (False, True):
    plan = zafx.stft_plan(zafx.hamming(W), H, onesided=onesided)
    out = zafx.pinned_empty(plan.out_shape(clips, N), plan.out_dtype)
    in_b, out_b = plan.clip_bytes(N)
    for mb in (0, 16, 32, 64, 128, 256, 512, 100000):
        chunk = 0 if mb == 0 else max(1, (mb << 20) // (in_b + out_b))
        plan.run_host(x, N, out=out, chunk_clips=chunk)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            plan.run_host(x, N, out=out, chunk_clips=chunk)
            best = min(best, time.perf_counter() - t0)
        print(f"onesided={onesided} chunk {mb:6d} MB ({chunk:4d} clips): {best * 1e3:7.2f} ms  {clips * N / best / 1e6:8.1f} Msamples/s  "
              f"down {out.nbytes / best / 1e9:5.1f} GB/s", flush=True)
    if not onesided:
        xp = np.array(x)
        t0 = time.perf_counter()
        got = plan.run_host(xp, N)
        t1 = time.perf_counter()
        got2 = plan.run_host(xp, N, out=got)
        t2 = time.perf_counter()
        print(f"pageable in, fresh pageable out: {clips * N / (t1 - t0) / 1e6:.1f} Msamples/s; reused pageable out: {clips * N / (t2 - t1) / 1e6:.1f}")
    del out
