import os, sys
import numpy as np
sys.path.insert(0, "zaf-python_amd")
import zafx
B, n = 1024, 441000
x = np.random.default_rng(5).standard_normal((8, n)).astype(np.float32)
d_in = zafx.DeviceBuffer.from_host(np.tile(x, (B // 8, 1)))
for hop in (4096, 2048):
    fb = zafx.melfilterbank(44100, 8192, 128)
    for ncoef in (None, 20):
        plan = zafx.mel_plan(zafx.hamming(8192), hop, fb, ncoef)
        d_out = zafx.DeviceBuffer(plan.out_shape(B, n), plan.out_dtype)
        for _ in range(3): plan.execute(d_in, d_out, B, n)
        plan.sync(); plan.timer_start()
        for _ in range(10): plan.execute(d_in, d_out, B, n)
        print("hop", hop, "mfcc" if ncoef else "mel", plan.last_kernel, round(plan.timer_stop() / 10, 3), "ms", flush=True)
        d_out.free()
