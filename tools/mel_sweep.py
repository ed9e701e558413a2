import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "zaf-python_amd"))
import zafx
B, N = 256, 441000
x = np.random.default_rng(0).standard_normal((8, N)).astype(np.float32)
d_x = zafx.DeviceBuffer.from_host(np.tile(x, (B // 8, 1)))
for wl in (512, 1024, 2048):
    fb = zafx.melfilterbank(44100, wl, 128 if wl >= 1024 else 64)
    for nc in (None, 20):
        plan = zafx.mel_plan(zafx.hamming(wl), wl // 2, fb, nc)
        d_out = zafx.DeviceBuffer(plan.out_shape(B, N), plan.out_dtype)
        plan.execute(d_x, d_out, B, N); plan.sync()
        plan.timer_start()
        for _ in range(10): plan.execute(d_x, d_out, B, N)
        ms = plan.timer_stop() / 10
        print(f"W={wl} {'mfcc' if nc else 'mel '} {ms:.3f} ms per 256 clips  ({B*N/ms/1e3:.0f} Msamples/s)")
