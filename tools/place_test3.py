#!/usr/bin/env python3
"""Several allocations of the STFT's input and output held at once: the rate for every (input, output) pair.
   gpurun -- 'python tools/place_test3.py'"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "zaf-python_amd"))
import zafx
B, N, W, H = 1024, 441000, 2048, 1024
K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
x = np.random.default_rng(0).standard_normal((8, N)).astype(np.float32)
host = np.tile(x, (B // 8, 1))
plan = zafx.stft_plan(zafx.hamming(W), H)
F, T = plan.out_dims(N)
ins, outs = [], []
for k in range(K):   # interleaved, as separate allocations
    ins.append(zafx.DeviceBuffer((B, N), np.float32))
    outs.append(zafx.DeviceBuffer((B, F, T), np.complex64))
for d in ins:
    d.upload(host)
def t(d_x, d_o, reps=12):
    for _ in range(4): plan.execute(d_x, d_o, B, N)
    plan.sync()
    t0 = time.perf_counter()
    for _ in range(reps): plan.execute(d_x, d_o, B, N)
    plan.sync()
    return (time.perf_counter() - t0) / reps * 1e3
for _ in range(30): plan.execute(ins[0], outs[0], B, N)   # clocks up
plan.sync()
print("rows: input k, columns: output k (ms per launch)")
for i in range(K):
    print(f"in {i} {ins[i].ptr.value:#x}: " + " ".join(f"{t(ins[i], outs[j]):.3f}" for j in range(K)))
print("outputs:", " ".join(f"{o.ptr.value:#x}" for o in outs))
