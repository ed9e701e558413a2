import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "zaf-python_amd"))
import zafx
mode = sys.argv[1]
B, N, W, H = 1024, 441000, 2048, 1024
x = np.random.default_rng(0).standard_normal((8, N)).astype(np.float32)
plan = zafx.stft_plan(zafx.hamming(W), H)
F, T = plan.out_dims(N)
host = np.tile(x, (B // 8, 1))
def alloc_in(): return zafx.DeviceBuffer((B, N), np.float32)
def alloc_out(): return zafx.DeviceBuffer((B, F, T), np.complex64)
if mode == "in_first": d_x = alloc_in(); d_o = alloc_out()
elif mode == "out_first": d_o = alloc_out(); d_x = alloc_in()
elif mode == "gap": d_x = alloc_in(); g = zafx.DeviceBuffer((1 << 30,), np.uint8); d_o = alloc_out()
elif mode == "realloc":   # allocate, free, allocate again (what a second bench kind sees)
    a = alloc_in(); b = alloc_out(); a.free(); b.free(); zafx.DeviceBuffer.drain_pool(); d_x = alloc_in(); d_o = alloc_out()
elif mode == "realloc_out": d_x = alloc_in(); b = alloc_out(); b.free(); zafx.DeviceBuffer.drain_pool(); d_o = alloc_out()
elif mode == "realloc_in": a = alloc_in(); a.free(); zafx.DeviceBuffer.drain_pool(); d_x = alloc_in(); d_o = alloc_out()
elif mode == "dummy_big":   # one 9 GiB allocation made and freed before the real ones
    g = zafx.DeviceBuffer((9 << 30,), np.uint8); g.free(); zafx.DeviceBuffer.drain_pool(); d_x = alloc_in(); d_o = alloc_out()
elif mode == "realloc_touched":   # the first pair is used before it is freed
    a = alloc_in(); b = alloc_out(); a.upload(host); plan.execute(a, b, B, N); plan.sync(); a.free(); b.free(); zafx.DeviceBuffer.drain_pool(); d_x = alloc_in(); d_o = alloc_out()
elif mode == "realloc3":
    for _ in range(3):
        a = alloc_in(); b = alloc_out(); a.free(); b.free(); zafx.DeviceBuffer.drain_pool()
    d_x = alloc_in(); d_o = alloc_out()
d_x.upload(host)
res = []
for rep in range(3):
    for _ in range(20): plan.execute(d_x, d_o, B, N)
    plan.sync()
    t0 = time.perf_counter()
    for _ in range(60): plan.execute(d_x, d_o, B, N)
    plan.sync()
    res.append((time.perf_counter() - t0) / 60 * 1e3)
print(mode, f"in {d_x.ptr.value:#x} out {d_o.ptr.value:#x}", " ".join(f"{r:.4f}" for r in res))
