"""Random geometries of every transform against the oracle -- run by tests/test_gpu_stress.py with fixed seeds, or by hand:  python tests/stress_random.py [seed [iterations]]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zaf-python_amd")); sys.path.insert(0, ROOT)
import numpy as np, zafx
from oracle import zaf_oracle as orc
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 12345
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 70
rng = np.random.default_rng(seed)
def relerr(a, b): return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30)) if a.size else float(a.shape != b.shape)
bad = 0
for it in range(iters):
    wl = int(2 ** rng.integers(7, 14))            # 128..8192
    if rng.integers(0, 4) == 0:
        wl = 2 * int(rng.integers(20, 1500))      # even, not a power of two (Bluestein forms); mdct needs an even window
    hop = int(rng.choice([wl // 2, wl // 4, wl // 2 + 2 * int(rng.integers(0, 20)), int(rng.integers(1, wl))]))
    n = int(rng.integers(1, 90000))
    nb = int(rng.integers(1, 7))
    x = rng.standard_normal((nb, n)).astype(np.float32)
    w = orc.hamming_periodic(wl)
    c = int(rng.integers(0, nb))
    try:
        s = zafx.stft_batch(x, w, hop)
        ref = orc.stft(x[c].astype(np.float64), w, hop)
        e1 = relerr(s[c], ref)
        e2 = 0.0
        if hop <= wl:
            y = zafx.istft_batch(s, w, hop)
            e2 = relerr(y[c], orc.istft(ref, w, hop))
        ws = orc.sine_window(wl)
        m = zafx.mdct_batch(x, ws)
        mref = orc.mdct(x[c].astype(np.float64), ws)
        e3 = relerr(m[c], mref)
        yi = zafx.imdct_batch(m, ws)
        e4 = relerr(yi[c], orc.imdct(mref, ws))
        e5 = 0.0
        if wl >= 256 and wl <= 2048 and (wl & (wl - 1)) == 0:
            fb = zafx.melfilterbank(44100, wl, int(rng.choice([40, 64, 128])))
            mm = zafx.mfcc_batch(x, w, hop, fb, 13) if hasattr(zafx, "mfcc_batch") else None
            if mm is not None:
                e5 = relerr(mm[c], orc.mfcc(x[c].astype(np.float64), w, hop, fb, 13))
        ok = e1 <= 1e-5 and e2 <= 3e-5 and e3 <= 1e-5 and e4 <= 3e-5 and e5 <= 1e-4
        if not ok:
            bad += 1
            print("FAIL", wl, hop, n, nb, c, e1, e2, e3, e4, e5)
    except Exception as exc:
        bad += 1
        print("EXC", wl, hop, n, nb, repr(exc)[:200])
print("seed", seed, "iterations", iters, "done, failures:", bad)
