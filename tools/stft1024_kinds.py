import sys, os, numpy as np
sys.path.insert(0, "zaf-python_amd")
import zafx
B, N, W = 1024, 441000, 1024
x = np.random.default_rng(0).standard_normal((8, N)).astype(np.float32)
d_x = zafx.DeviceBuffer.from_host(np.tile(x, (B // 8, 1)))
for one in (False, True, "magnitude"):
    for al in (0, 16 if one != "magnitude" else 32):
        pl = zafx.stft_plan(zafx.hamming(W), W // 2, onesided=one, row_align=al)
        d = zafx.DeviceBuffer(pl.out_shape(B, N), pl.out_dtype)
        for _ in range(30): pl.execute(d_x, d, B, N)
        pl.sync(); pl.timer_start()
        for _ in range(20): pl.execute(d_x, d, B, N)
        print(one, al, round(pl.timer_stop() / 20, 3), pl.last_kernel, flush=True)
        d.free()
