"""Round 5: device-resident int16 PCM straight into the transform (Plan.execute_pcm) against float32 samples and against the pre-pass."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "zaf-python_amd"))
import zafx
B, N = 1024, 441000
w = zafx.hamming(2048)
fb = zafx.melfilterbank(44100, 2048, 128)
rng = np.random.default_rng(0)
base = rng.integers(-32768, 32767, size=(8, N, 2), endpoint=True).astype(np.int16)
def timed(fn, plan, reps=20):
    for _ in range(5): fn()
    plan.sync(); plan.timer_start()
    for _ in range(reps): fn()
    return plan.timer_stop() / reps
for name, plan in (("stft", zafx.stft_plan(w, 1024)), ("stft1", zafx.stft_plan(w, 1024, onesided=True)), ("mel", zafx.mel_plan(w, 1024, fb)), ("mfcc", zafx.mel_plan(w, 1024, fb, 20)), ("|X|", zafx.stft_plan(w, 1024, onesided="magnitude")), ("mdct", zafx.mdct_plan(zafx.kaiser_bessel_derived(2048))), ("mdct T=433", zafx.mdct_plan(zafx.kaiser_bessel_derived(2048)))):
    if name == "mdct T=433":
        N = 442024
        base = rng.integers(-32768, 32767, size=(8, N, 2), endpoint=True).astype(np.int16)
    d_out = zafx.DeviceBuffer(plan.out_shape(B, N), plan.out_dtype)
    d_x = zafx.DeviceBuffer((B, N), np.float32)
    for ch in (1, 2):
        pcm = np.tile(np.ascontiguousarray(base[:, :, :ch]), (B // 8, 1, 1))
        d_pcm = zafx.DeviceBuffer.from_host(pcm)
        plan.pcm_to_float(d_pcm, d_x, B, N, ch); plan.sync()
        t_f32 = timed(lambda: plan.execute(d_x, d_out, B, N), plan)
        t_two = timed(lambda: (plan.pcm_to_float(d_pcm, d_x, B, N, ch), plan.execute(d_x, d_out, B, N)), plan)
        t_pcm = timed(lambda: plan.execute_pcm(d_pcm, d_out, B, N, ch), plan)
        print(f"{name} int16 x {ch} channel(s): float32 samples {t_f32:.3f} ms | pre-pass + transform {t_two:.3f} ms | int16 in the loads {t_pcm:.3f} ms ({plan.last_kernel}) = {B * N / t_pcm / 1e6:.1f} Gsamples/s", flush=True)
        d_pcm.free()
    d_out.free(); d_x.free()
