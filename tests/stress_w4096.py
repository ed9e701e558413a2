"""Random geometries at W = 4096 (the band kernels of round 3: k_stft_ft16b / bc, k_istft_ft16b and -- hop W / 2, round 6 -- k_istft_ft16d, k_mdct_ft32b, the 16-frame k_imdct, k_melfb) and at
W = 8192 (two draws in five; round 5: k_stft_ft16q, k_mdct_ft32q, mel / mfcc through k_stft_ft16q + k_melfb; round 6: k_istft_ft8q at hop W / 2, k_imdct_q) against the oracle -- run by tests/test_gpu_stress.py with fixed seeds, or by hand:
    python tests/stress_w4096.py [seed [iterations]]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zaf-python_amd")); sys.path.insert(0, ROOT)
import numpy as np, zafx
from oracle import zaf_oracle as orc
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rng = np.random.default_rng(seed)
def relerr(a, b): return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30)) if a.size else float(a.shape != b.shape)
bad = 0
for it in range(iters):
    wl = 4096 if rng.integers(0, 5) >= 2 else 8192
    hop = int(rng.choice([wl // 2, wl // 4, wl // 8, 4 * int(rng.integers(128, 1100)), int(rng.integers(512, wl + 1)), 2 * int(rng.integers(300, 2000))]))
    n = int(rng.choice([int(rng.integers(1, 200000)), 4 * int(rng.integers(1, 50000)), 2048 * 16 * int(rng.integers(1, 5))]))
    nb = int(rng.choice([1, 2, 3, 5, 40, 300])) if n < 60000 else int(rng.integers(1, 5))
    x = rng.standard_normal((nb, n)).astype(np.float32)
    w = orc.hamming_periodic(wl)
    c = int(rng.integers(0, nb))
    x64 = x[c].astype(np.float64)
    try:
        ref = orc.stft(x64, w, hop)
        half = wl // 2 + 1
        e = [relerr(zafx.stft_batch(x, w, hop)[c], ref),
             relerr(zafx.stft_batch(x, w, hop, onesided=True)[c], ref[:half]),
             relerr(zafx.stft_batch(x, w, hop, onesided="magnitude")[c], np.abs(ref[:half]))]
        if hop <= wl:
            spec = np.stack([orc.stft(x[i].astype(np.float64), w, hop) for i in {c, 0}])
            spec = spec + 0.03 * (rng.standard_normal(spec.shape) + 1j * rng.standard_normal(spec.shape))
            e.append(relerr(zafx.istft_batch(spec, w, hop)[0], orc.istft(spec[0], w, hop)) / 3)
            e.append(relerr(zafx.istft_batch(ref[None, :half], w, hop, onesided=True)[0], orc.istft(ref, w, hop)) / 3)
        ws = orc.sine_window(wl)
        m = zafx.mdct_batch(x, ws)
        mref = orc.mdct(x64, ws)
        e.append(relerr(m[c], mref))
        e.append(relerr(zafx.imdct_batch(m, ws)[c], orc.imdct(mref, ws)) / 3)
        fb = zafx.melfilterbank(44100, wl, int(rng.choice([40, 128, 256])))
        e.append(relerr(zafx.melspectrogram_batch(x, w, hop, fb)[c], orc.melspectrogram(x64, w, hop, fb)) / 10)
        e.append(relerr(zafx.mfcc_batch(x, w, hop, fb, 13)[c], orc.mfcc(x64, w, hop, fb, 13)) / 10)
        if not max(e) <= 1e-5:
            bad += 1
            print("FAIL", wl, hop, n, nb, c, ["%.1e" % v for v in e])
    except Exception as exc:
        bad += 1
        print("EXC", wl, hop, n, nb, repr(exc)[:300])
print("seed", seed, "iterations", iters, "done, failures:", bad)
