"""Random geometries of the variants `stress_random.py` leaves out -- frame-major layout, one-sided / magnitude / power spectra,
float64 mode, mel without DCT, CQT spectrogram / chromagram with random kernels, DCT / DST types 1-4, PCM ingest -- against the
oracle.  Run by tests/test_gpu_stress.py with a fixed seed, or by hand:  python tests/stress_random_more.py [seed [iterations]]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zaf-python_amd"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import zafx  # noqa: E402
from oracle import zaf_oracle as orc  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 777
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(seed)


def relerr(a, b):
    if a.shape != b.shape:
        return float("inf")
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30)) if a.size else 0.0


bad = 0


def check(name, err, tol, *ctx):
    global bad
    if not err <= tol:
        bad += 1
        print("FAIL", name, err, "tol", tol, *ctx)


for it in range(iters):
    wl = int(2 ** rng.integers(6, 13))   # 64..4096
    hop = int(rng.choice([wl // 2, wl // 4, int(rng.integers(1, wl + 1))]))
    n = int(rng.integers(1, 60000))
    nb = int(rng.integers(1, 5))
    x = rng.standard_normal((nb, n)).astype(np.float32)
    c = int(rng.integers(0, nb))
    w = orc.hamming_periodic(wl)
    ctx = (wl, hop, n, nb, c)
    try:
        ref = orc.stft(x[c].astype(np.float64), w, hop)
        # frame-major layout, spectrum kinds
        s_tf = zafx.stft_batch(x, w, hop, layout="TF")
        check("stft TF", relerr(s_tf[c].T, ref), 1e-5, *ctx)
        s1 = zafx.stft_batch(x, w, hop, onesided=True)
        check("stft onesided", relerr(s1[c], ref[: wl // 2 + 1]), 1e-5, *ctx)
        sm = zafx.stft_batch(x, w, hop, onesided="magnitude", layout="TF")
        check("stft magnitude TF", relerr(sm[c].T, np.abs(ref[: wl // 2 + 1])), 1e-5, *ctx)
        sp = zafx.stft_batch(x, w, hop, onesided="power")
        check("stft power", relerr(sp[c], np.abs(ref[: wl // 2 + 1]) ** 2), 2e-5, *ctx)
        yref = orc.istft(ref, w, hop)
        y_tf = zafx.istft_batch(s_tf, w, hop, layout="TF")
        check("istft TF", relerr(y_tf[c], yref), 3e-5, *ctx)
        y1 = zafx.istft_batch(s1, w, hop, onesided=True)
        check("istft onesided", relerr(y1[c], yref), 3e-5, *ctx)
        # float64 mode
        s64 = zafx.stft_batch(x.astype(np.float64), w, hop, f64=True)
        check("stft f64", relerr(s64[c], ref), 1e-12, *ctx)
        y64 = zafx.istft_batch(s64, w, hop, f64=True)
        check("istft f64", relerr(y64[c], yref), 1e-12, *ctx)
        ws = orc.sine_window(wl)
        mref = orc.mdct(x[c].astype(np.float64), ws)
        m_tf = zafx.mdct_batch(x, ws, layout="TF")
        check("mdct TF", relerr(m_tf[c].T, mref), 1e-5, *ctx)
        check("imdct TF", relerr(zafx.imdct_batch(m_tf, ws, layout="TF")[c], orc.imdct(mref, ws)), 3e-5, *ctx)
        m64 = zafx.mdct_batch(x.astype(np.float64), ws, f64=True)
        check("mdct f64", relerr(m64[c], mref), 1e-12, *ctx)
        check("imdct f64", relerr(zafx.imdct_batch(m64, ws, f64=True)[c], orc.imdct(mref, ws)), 1e-12, *ctx)
        if 256 <= wl <= 2048:
            nf = int(rng.choice([20, 40, 64, 100, 128]))
            fb = zafx.melfilterbank(44100, wl, nf)
            mel = zafx.melspectrogram_batch(x, w, hop, fb, layout=str(rng.choice(["FT", "TF"])))
            mel_ref = orc.melspectrogram(x[c].astype(np.float64), w, hop, fb)
            got = mel[c] if mel.shape[1:] == mel_ref.shape else mel[c].T
            check("mel", relerr(got, mel_ref), 1e-4, *ctx, nf)
            nc = int(rng.integers(1, min(nf, 40)))
            mf = zafx.mfcc_batch(x, w, hop, fb, nc, layout="TF")
            check("mfcc TF", relerr(mf[c].T, orc.mfcc(x[c].astype(np.float64), w, hop, fb, nc)), 1e-4, *ctx, nf, nc)
    except Exception as exc:   # noqa: BLE001
        bad += 1
        print("EXC", *ctx, repr(exc)[:300])

# CQT with random kernels (fft lengths 2048 .. 32768), spectrogram and chromagram
for it in range(max(4, iters // 6)):
    fs = int(rng.choice([8000, 16000, 22050, 44100]))
    res = int(rng.choice([2, 6, 12, 24]))
    fmin = float(rng.choice([55.0, 110.0, 220.0, 32.7]))
    octaves = int(rng.integers(1, 6))
    fmax = min(fmin * 2 ** octaves, fs / 2 * 0.9)
    tres = int(rng.choice([10, 25, 50, 100]))
    n = int(rng.integers(fs // 4, 3 * fs))
    nb = int(rng.integers(1, 4))
    x = rng.standard_normal((nb, n)).astype(np.float32)
    c = int(rng.integers(0, nb))
    ctx = (fs, res, fmin, fmax, tres, n, nb, c)
    try:
        kern = zafx.cqtkernel(fs, res, fmin, fmax)
        if round(fs / tres) > kern.shape[1]:   # negative padding in the reference (np.pad raises): ValueError here as well
            try:
                zafx.cqtspectrogram_batch(x, fs, tres, kern)
                check("cqt step > fft_length must raise", 1.0, 0.0, *ctx)
            except ValueError:
                pass
            continue
        layout = str(rng.choice(["FT", "TF"]))
        got = zafx.cqtspectrogram_batch(x, fs, tres, kern, layout=layout)[c]
        ref = orc.cqtspectrogram(x[c].astype(np.float64), fs, tres, kern)
        check("cqt " + layout, relerr(got if layout == "FT" else got.T, ref), 1e-4, *ctx)
        got = zafx.cqtchromagram_batch(x, fs, tres, res, kern)[c]
        check("chroma", relerr(got, orc.cqtchromagram(x[c].astype(np.float64), fs, tres, res, kern)), 1e-4, *ctx)
    except Exception as exc:   # noqa: BLE001
        bad += 1
        print("EXC cqt", *ctx, repr(exc)[:300])

# DCT / DST types 1-4, random lengths
for it in range(max(8, iters // 3)):
    n = int(rng.choice([int(rng.integers(2, 3000)), 4 * int(rng.integers(1, 2049)), int(rng.integers(2, 8193))]))   # (lengths 4 j: k_dct_bsh for types 2-4)
    nb = int(rng.integers(1, 40)) if n < 3000 else int(rng.integers(1, 600))   # (more rows than resident workgroups now and then)
    kind = int(rng.integers(1, 5))
    x = rng.standard_normal((nb, n)).astype(np.float32)
    c = int(rng.integers(0, nb))
    try:
        check("dct", relerr(zafx.dct_batch(x, kind)[c], orc.dct(x[c].astype(np.float64), kind)), 1e-5, n, nb, kind)
        check("dst", relerr(zafx.dst_batch(x, kind)[c], orc.dst(x[c].astype(np.float64), kind)), 1e-5, n, nb, kind)
    except Exception as exc:   # noqa: BLE001
        bad += 1
        print("EXC dct/dst", n, nb, kind, repr(exc)[:300])

# PCM ingest: int16 / int32, 1-3 channels
for it in range(6):
    nb, n, ch = int(rng.integers(1, 5)), int(rng.integers(1, 50000)), int(rng.integers(1, 4))
    dt = np.int16 if rng.integers(0, 2) else np.int32
    info = np.iinfo(dt)
    pcm = rng.integers(info.min, info.max, size=(nb, n, ch), endpoint=True).astype(dt)
    want = (pcm.astype(np.float64) / 2.0 ** (8 * pcm.itemsize - 1)).mean(axis=2)
    try:
        check("pcm", float(np.max(np.abs(zafx.pcm_to_mono(pcm) - want))), 2e-7, nb, n, ch, dt.__name__)
    except Exception as exc:   # noqa: BLE001
        bad += 1
        print("EXC pcm", nb, n, ch, repr(exc)[:300])

print("seed", seed, "iterations", iters, "done, failures:", bad)
