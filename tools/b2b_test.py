import sys, time
sys.path.insert(0, "zaf-python_amd")
import numpy as np, zafx, ctypes
from zafx import _lib
B, N, W = 1024, 441000, 2048
x = np.random.default_rng(0).standard_normal((8, N)).astype(np.float32)
d_x = zafx.DeviceBuffer.from_host(np.tile(x, (B // 8, 1)))
w = zafx.kaiser_bessel_derived(W)
plan = zafx.mdct_plan(w)
F, T = plan.out_dims(N)
d_o = zafx.DeviceBuffer(plan.out_shape(B, N), np.float32)
for _ in range(3): plan.execute(d_x, d_o, B, N)
plan.sync()
# isolated
iso = []
for _ in range(10):
    plan.timer_start(); plan.execute(d_x, d_o, B, N); iso.append(plan.timer_stop())
# back to back, K launches as a block, several K
for K in (2, 5, 20, 100, 400):
    plan.sync(); time.sleep(0.2)
    plan.timer_start()
    for _ in range(K): plan.execute(d_x, d_o, B, N)
    ms = plan.timer_stop() / K
    print("K", K, "ms/launch", round(ms, 4))
print("isolated median", round(float(np.median(iso)), 4), "min", round(min(iso), 4))
