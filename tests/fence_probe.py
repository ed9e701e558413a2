"""Helper of tests/test_gpu_fence.py: run a fixed set of transforms with whatever libzafx build ZAFX_LIBRARY names and
save the raw outputs (compared bit for bit between the default and the -DZAFX_WAVE_SYNC_FENCE build)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zaf-python_amd"))
import zafx  # noqa: E402


def clip(c, n):
    return np.random.default_rng([77, c]).standard_normal(n).astype(np.float32)


def main(out_path):
    res = {}
    # every FFT size whose frame is owned by one wavefront (the unfenced exchange): W = 64 ... 2048, both layouts
    for wl, hop, n in [(64, 32, 1000), (128, 64, 3000), (256, 100, 5000), (512, 256, 9000), (1024, 512, 20000), (2048, 1024, 50000),
                       (2048, 512, 30001), (4096, 2048, 40000), (8192, 4096, 4096 * 31 - 9)]:   # (W = 8192: 32 frames, k_stft_ft16q / k_mdct_ft32q)
        x = np.stack([clip(c, n) for c in range(3)])
        w = zafx.hamming(wl)
        for layout in ("FT", "TF"):
            s = zafx.stft_batch(x, w, hop, layout=layout)
            res[f"stft_{wl}_{hop}_{layout}"] = s
            res[f"istft_{wl}_{hop}_{layout}"] = zafx.istft_batch(s, w, hop, layout=layout)
        res[f"mag_{wl}_{hop}"] = zafx.stft_batch(x, w, hop, onesided="magnitude")
        kbd = zafx.kaiser_bessel_derived(wl)
        for layout in ("FT", "TF"):
            m = zafx.mdct_batch(x, kbd, layout=layout)
            res[f"mdct_{wl}_{layout}"] = m
            res[f"imdct_{wl}_{layout}"] = zafx.imdct_batch(m, kbd, layout=layout)
        if wl <= 2048:
            fb = zafx.melfilterbank(44100, wl, 40 if wl >= 256 else 8)
            res[f"mel_{wl}"] = zafx.melspectrogram_batch(x, w, hop, fb)
            res[f"mfcc_{wl}"] = zafx.mfcc_batch(x, w, hop, fb, 5)
    # the float64 kernels of the tiled structure (a frame per wavefront, the same unfenced exchange on complex128 / float64 frames)
    x64 = np.stack([clip(c, 1024 * 37 + 5) for c in range(3)]).astype(np.float64)
    w, kbd = zafx.hamming(2048), zafx.kaiser_bessel_derived(2048)
    s64 = zafx.stft_batch(x64, w, 1024, f64=True)
    m64 = zafx.mdct_batch(x64, kbd, f64=True)
    res["stft64"], res["mdct64"] = s64.view(np.float64), m64
    res["istft64"], res["imdct64"] = zafx.istft_batch(s64, w, 1024, f64=True), zafx.imdct_batch(m64, kbd, f64=True)
    x = np.stack([clip(c, 150000) for c in range(2)])
    for fmin, fmax in ((55, 3520), (220, 1760), (880, 3520)):   # fft_length 32768 (16 x 1024 split), 8192, 2048
        ck = zafx.cqtkernel(44100, 24, fmin, fmax)
        res[f"cqt_{fmin}"] = zafx.cqtspectrogram_batch(x, 44100, 25, ck)
    np.savez(out_path, **res)
    print(zafx.library_path(), len(res))


if __name__ == "__main__":
    main(sys.argv[1])
