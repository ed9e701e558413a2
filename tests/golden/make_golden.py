#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REAL reference.

Run in the build container only (it needs /root/reference/zaf.py, which never
travels to the GPU box):

    MPLBACKEND=Agg python tests/golden/make_golden.py

What is written (all .npz, float64/complex128, < 1 MB each):
  tiny.npz      full inputs + full outputs of all ten functions at W=64, hop=32,
                N in {1, 63, 64, 65, 1000}, plus hop=16 (75 % overlap) cases
  consts.npz    melfilterbank(44100,2048,128), melfilterbank(44100,2048,40),
                cqtkernel(44100,24,55,3520) and a small cqtkernel, as CSR triplets
  config.npz    BASELINE.json config-size cases (10 s / 30 s clips): input checksum,
                output shape, 4096 random probes (index + value), row sums, column
                sums and L2 norm per function
  lengths.npz   frame-count / output-length table of SURVEY.md section 4
  dctdst.npz    zaf.dct / zaf.dst, types 1-4, lengths 8, 9, 100, 1024, 63, 64, 65, 1023, 1025
  signals.npz   signals that are not white noise (tests/signals.py: silence, DC, on-bin / between-bin full-scale sines, two
                tones 100 dB apart, impulses on hop boundaries, a chirp, noise at -90 dBFS, clipped int16 PCM): rows
                0..W/2 of zaf.stft (the mirror rows are their conjugates to 1e-15 of the peak, checked here), istft, melspectrogram, mfcc,
                mdct, imdct at W = 2048 / hop 1024 on 3072 samples; cqtspectrogram / cqtchromagram on 17640 samples
  cqtfull.npz   the kernel of cqtkernel's own docstring example (zaf.py:476-483: 55 Hz ... fs/2, 208 bins, 60 879
                non-zeros, columns on both halves of the spectrum): nnz per row, column range, value probes, and the
                full cqtspectrogram / cqtchromagram of a 100 000-sample clip with it

The fixtures are DATA (inputs and expected outputs); no reference source text
is stored.
"""
import os
import sys

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import scipy.signal.windows  # noqa: E402
import zaf  # noqa: E402  (the reference)

HERE = os.path.dirname(os.path.abspath(__file__))


def clip(seed, c, n):
    """Synthetic input of SURVEY 8(d): white Gaussian noise, f32-rounded."""
    return np.random.default_rng([seed, c]).standard_normal(n).astype(np.float32)


def csr_triplet(m):
    m = m.tocsr()
    m.sort_indices()
    return m.data, m.indices.astype(np.int64), m.indptr.astype(np.int64), np.array(m.shape, dtype=np.int64)


def probes(arr, rng, k=4096):
    flat = arr.reshape(-1)
    k = min(k, flat.size)
    idx = rng.choice(flat.size, size=k, replace=False).astype(np.int64)
    out = {"shape": np.array(arr.shape, dtype=np.int64), "idx": idx, "val": flat[idx]}
    if arr.ndim == 2:
        out["rowsum"] = arr.sum(axis=1)
        out["colsum"] = arr.sum(axis=0)
    out["l2"] = np.array(np.sqrt(np.sum(np.abs(arr) ** 2)))
    out["maxabs"] = np.array(np.max(np.abs(arr)))
    return out


def make_tiny():
    out = {}
    wl = 64
    ham = scipy.signal.windows.hamming(wl, sym=False)
    sine = np.sin(np.pi / wl * (np.arange(wl) + 0.5))
    kbd = scipy.signal.windows.kaiser_bessel_derived(wl, beta=5 * np.pi)
    fb = zaf.melfilterbank(8000, wl, 8)
    ck = zaf.cqtkernel(4000, 12, 200, 1600)
    out["ham"], out["sine"], out["kbd"] = ham, sine, kbd
    out["fb_dense"] = fb.toarray()
    out["ck_dense"] = ck.toarray()
    for n in (1, 63, 64, 65, 1000):
        x = clip(7, n, n).astype(np.float64)
        out[f"x_{n}"] = x
        for hop in (32, 16):
            s = zaf.stft(x, ham, hop)
            out[f"stft_{n}_{hop}"] = s
            out[f"istft_{n}_{hop}"] = zaf.istft(s, ham, hop)
            out[f"mel_{n}_{hop}"] = zaf.melspectrogram(x, ham, hop, fb)
            out[f"mfcc_{n}_{hop}"] = zaf.mfcc(x, ham, hop, fb, 5)
        for name, w in (("sine", sine), ("kbd", kbd)):
            m = zaf.mdct(x, w)
            out[f"mdct_{name}_{n}"] = m
            out[f"imdct_{name}_{n}"] = zaf.imdct(m, w)
    # non-Hermitian spectrum through istft (reference takes real(ifft) of anything)
    rng = np.random.default_rng(11)
    z = rng.standard_normal((wl, 9)) + 1j * rng.standard_normal((wl, 9))
    out["istft_generic_in"] = z
    out["istft_generic_out"] = zaf.istft(z, ham, 32)
    # CQT tiny: fs 4000, 12 bins/octave, 200..1600 Hz -> 36 bins, fft_len 512
    for n in (400, 4000, 4321):
        x = clip(9, n, n).astype(np.float64)
        out[f"xq_{n}"] = x
        out[f"cqt_{n}"] = zaf.cqtspectrogram(x, 4000, 50, ck)
        out[f"chroma_{n}"] = zaf.cqtchromagram(x, 4000, 50, 12, ck)
    np.savez_compressed(os.path.join(HERE, "tiny.npz"), **out)


def make_consts():
    out = {}
    for tag, m in (
        ("fb128", zaf.melfilterbank(44100, 2048, 128)),
        ("fb40", zaf.melfilterbank(44100, 2048, 40)),
        ("ck", zaf.cqtkernel(44100, 24, 55, 3520)),
        ("ck_small", zaf.cqtkernel(4000, 12, 200, 1600)),
    ):
        d, i, p, s = csr_triplet(m)
        out[f"{tag}_data"], out[f"{tag}_indices"], out[f"{tag}_indptr"], out[f"{tag}_shape"] = d, i, p, s
    np.savez_compressed(os.path.join(HERE, "consts.npz"), **out)


def make_config():
    out = {}
    rng = np.random.default_rng(2024)
    wl, hop, n = 2048, 1024, 441000
    ham = scipy.signal.windows.hamming(wl, sym=False)
    kbd = scipy.signal.windows.kaiser_bessel_derived(wl, beta=5 * np.pi)
    fb = zaf.melfilterbank(44100, wl, 128)
    for c in (0, 1):
        x = clip(0, c, n).astype(np.float64)
        out[f"S{c}_x_sum"] = np.array(x.sum())
        out[f"S{c}_x_head"] = x[:16]
        s = zaf.stft(x, ham, hop)
        funcs = {
            "stft": s,
            "istft": zaf.istft(s, ham, hop),
            "mel": zaf.melspectrogram(x, ham, hop, fb),
            "mfcc": zaf.mfcc(x, ham, hop, fb, 20),
        }
        m = zaf.mdct(x, kbd)
        funcs["mdct"] = m
        funcs["imdct"] = zaf.imdct(m, kbd)
        for name, arr in funcs.items():
            for k, v in probes(arr, rng).items():
                out[f"S{c}_{name}_{k}"] = v
    # CQT config Q: 30 s clip, 24 bins/octave, 55..3520 Hz, 25 frames/s
    ck = zaf.cqtkernel(44100, 24, 55, 3520)
    nq = 1323000
    x = clip(0, 0, nq).astype(np.float64)
    out["Q0_x_sum"] = np.array(x.sum())
    out["Q0_x_head"] = x[:16]
    q = zaf.cqtspectrogram(x, 44100, 25, ck)
    for k, v in probes(q, rng).items():
        out[f"Q0_cqt_{k}"] = v
    ch = zaf.cqtchromagram(x, 44100, 25, 24, ck)
    for k, v in probes(ch, rng).items():
        out[f"Q0_chroma_{k}"] = v
    np.savez_compressed(os.path.join(HERE, "config.npz"), **out)


def make_lengths():
    wl, hop = 2048, 1024
    ham = scipy.signal.windows.hamming(wl, sym=False)
    kbd = scipy.signal.windows.kaiser_bessel_derived(wl, beta=5 * np.pi)
    ns = np.array([1, 1023, 1024, 1025, 2047, 2048, 2049, 5000, 441000], dtype=np.int64)
    rows = []
    for n in ns:
        x = clip(3, int(n), int(n)).astype(np.float64)
        s = zaf.stft(x, ham, hop)
        m = zaf.mdct(x, kbd)
        rows.append([n, s.shape[1], m.shape[1], len(zaf.istft(s, ham, hop)), len(zaf.imdct(m, kbd))])
    np.savez_compressed(os.path.join(HERE, "lengths.npz"), table=np.array(rows, dtype=np.int64))


def make_dctdst():
    """zaf.dct / zaf.dst types 1-4: full vectors, odd and power-of-two lengths (63 / 64 / 65 / 1023 / 1025:
    the lengths whose N/2, N-1 or N+1 is a power of two run on the FFT core, zafx_dct.hip)."""
    out = {}
    rng = np.random.default_rng(77)
    for n in (8, 9, 100, 1024, 63, 64, 65, 1023, 1025):
        x = rng.standard_normal(n).astype(np.float32).astype(np.float64)
        out[f"x_{n}"] = x
        for t in (1, 2, 3, 4):
            out[f"dct{t}_{n}"] = zaf.dct(x.copy(), t)
            out[f"dst{t}_{n}"] = zaf.dst(x.copy(), t)
    np.savez_compressed(os.path.join(HERE, "dctdst.npz"), **out)


def make_cqtfull():
    out = {}
    rng = np.random.default_rng(479)
    ck = zaf.cqtkernel(44100, 24, 55, 44100 / 2)   # zaf.py:476-483
    d, i, p, s = csr_triplet(ck)
    out["shape"], out["indptr"] = s, p
    out["col_min"], out["col_max"] = np.array(i.min()), np.array(i.max())
    idx = rng.choice(len(d), size=2048, replace=False).astype(np.int64)
    out["probe_idx"], out["probe_col"], out["probe_val"] = idx, i[idx], d[idx]
    x = clip(5, 0, 100000).astype(np.float64)
    out["cqt"] = zaf.cqtspectrogram(x, 44100, 25, ck)
    out["chroma"] = zaf.cqtchromagram(x, 44100, 25, 24, ck)
    np.savez_compressed(os.path.join(HERE, "cqtfull.npz"), **out)


def make_signals():
    """Verdict r4 item 2: one clip per signal of tests/signals.py through every function of the path at W = 2048."""
    sys.path.insert(0, os.path.dirname(HERE))
    import signals as sig
    out = {}
    wl, hop = sig.W, sig.HOP
    ham = scipy.signal.windows.hamming(wl, sym=False)
    kbd = scipy.signal.windows.kaiser_bessel_derived(wl, beta=5 * np.pi)
    fb = zaf.melfilterbank(sig.FS, wl, 128)
    ck = zaf.cqtkernel(sig.FS, 24, 55, 3520)
    for name in sig.NAMES:
        x = sig.signal(name, sig.N_FRAMES).astype(np.float64)
        out[f"{name}_x_sum"], out[f"{name}_x_abs"] = np.array(x.sum()), np.array(np.abs(x).sum())
        s = zaf.stft(x, ham, hop)
        # real input: the mirror rows are the conjugates up to the transform's own rounding (measured <= 2.2e-16 of the peak)
        assert np.abs(s[wl // 2 + 1:] - np.conj(s[wl // 2 - 1:0:-1])).max() <= 1e-15 * max(np.abs(s).max(), 1e-300)
        out[f"{name}_stft"] = s[: wl // 2 + 1]
        out[f"{name}_istft"] = zaf.istft(s, ham, hop)
        out[f"{name}_mel"] = zaf.melspectrogram(x, ham, hop, fb)
        out[f"{name}_mfcc"] = zaf.mfcc(x, ham, hop, fb, 20)
        m = zaf.mdct(x, kbd)
        out[f"{name}_mdct"] = m
        out[f"{name}_imdct"] = zaf.imdct(m, kbd)
        xq = sig.signal(name, sig.N_CQT).astype(np.float64)
        out[f"{name}_xq_sum"], out[f"{name}_xq_abs"] = np.array(xq.sum()), np.array(np.abs(xq).sum())
        out[f"{name}_cqt"] = zaf.cqtspectrogram(xq, sig.FS, 25, ck)
        out[f"{name}_chroma"] = zaf.cqtchromagram(xq, sig.FS, 25, 24, ck)
    np.savez_compressed(os.path.join(HERE, "signals.npz"), **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "signals":   # (added in round 5; the other files are unchanged)
        make_signals()
        print("signals.npz", os.path.getsize(os.path.join(HERE, "signals.npz")))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "cqtfull":   # (added in round 2; the other files are unchanged)
        make_cqtfull()
        print("cqtfull.npz", os.path.getsize(os.path.join(HERE, "cqtfull.npz")))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "dctdst":    # (extended in round 4: the first four lengths come out unchanged, same generator state)
        make_dctdst()
        print("dctdst.npz", os.path.getsize(os.path.join(HERE, "dctdst.npz")))
        sys.exit(0)
    make_tiny()
    make_consts()
    make_config()
    make_lengths()
    make_dctdst()
    make_cqtfull()
    make_signals()
    for f in ("tiny.npz", "consts.npz", "config.npz", "lengths.npz", "dctdst.npz", "cqtfull.npz", "signals.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)))
