import sys, os, numpy as np
sys.path.insert(0, "zaf-python_amd")
import zafx
W = 4096
ham = zafx.hamming(W)
for B, N, hop in ((1024, 441000, 1024), (512, 882000, 2048), (1024, 441000, 2048), (1024, 440320, 2048), (512, 441000, 1024), (1024, 441000, 512)):
    x = np.random.default_rng(0).standard_normal((8, N)).astype(np.float32)
    d_x = zafx.DeviceBuffer.from_host(np.tile(x, (B // 8, 1)))
    pl = zafx.stft_plan(ham, hop, row_align=16)
    F, T = pl.out_dims(N)
    d = zafx.DeviceBuffer(pl.out_shape(B, N), pl.out_dtype)
    for _ in range(30): pl.execute(d_x, d, B, N)
    pl.sync(); pl.timer_start()
    for _ in range(20): pl.execute(d_x, d, B, N)
    ms = pl.timer_stop() / 20
    tiles = B * ((T + 15) // 16)
    gb = B * (4 * N + 8 * W * T) / 1e9
    print(f"B={B} N={N} hop={hop} T={T}: {ms:.3f} ms, {ms*1e3/ (tiles/256):.1f} us per tile per CU, {gb/ms/8:.3f} of HBM ({pl.last_kernel})", flush=True)
    d.free(); d_x.free()
