import sys, os, time
sys.path.insert(0, "zaf-python_amd")
import numpy as np, zafx
B, N, W = 1024, 441000, 2048
x = np.random.default_rng(0).standard_normal((8, N)).astype(np.float32)
d_x = zafx.DeviceBuffer.from_host(np.tile(x, (B // 8, 1)))
for align in (0, 32):
    w = zafx.kaiser_bessel_derived(W)
    fwd = zafx.mdct_plan(w, row_align=align); inv = zafx.mdct_plan(w, inverse=True, row_align=align)
    F, T = fwd.out_dims(N)
    shp = fwd.out_shape(B, N)
    d_c = zafx.DeviceBuffer(shp, np.float32)
    d_y = zafx.DeviceBuffer((B, inv.out_dims(T)[0]), np.float32)
    for plan, a, b, n in ((fwd, d_x, d_c, N), (inv, d_c, d_y, T)):
        for _ in range(3): plan.execute(a, b, B, n)
        plan.sync()
        ts = []
        for _ in range(10):
            plan.timer_start(); plan.execute(a, b, B, n); ts.append(plan.timer_stop())
        print("align", align, plan.kernel_name, shp, "median ms", round(float(np.median(ts)), 4), "min", round(min(ts), 4))
