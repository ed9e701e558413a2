"""Deterministic test signals that are NOT white noise (verdict r4 item 2), shared by
tests/golden/make_golden.py (which feeds them to the real reference) and the parity tests.

The reference's own examples run on music (zaf.py:62-66); where a float32 pipeline can part from the
float64 reference is on near-silent bands under a loud one, exact zeros, and full-scale / clipped
input (zaf.py:443-446 takes the log of whatever the filterbank leaves).  Every recipe returns float32
(what both sides are fed); `clipped_pcm` also has an int16 form for the PCM entry points
(zaf.py:1202 scales by 2**15).
"""
import numpy as np

FS = 44100
W = 2048
HOP = 1024
N_FRAMES = 3072        # stft / mdct family: 4 STFT frames (two whole, two over the zero padding)
N_CQT = 17640          # cqt family: 10 frames of 32768 samples

NAMES = ("silence", "dc", "sine_bin", "sine_half", "two_tones", "impulse", "chirp", "noise_m90", "clipped_pcm")


def clipped_pcm16(n):
    """A sine 3.5 dB over full scale, rounded and clipped the way a 16-bit recorder does."""
    t = np.arange(n, dtype=np.float64)
    v = np.rint(1.5 * 32767.0 * np.sin(2 * np.pi * 97.3 * t / W))
    return np.clip(v, -32768, 32767).astype(np.int16)


def signal(name, n):
    t = np.arange(n, dtype=np.float64)
    if name == "silence":
        x = np.zeros(n)
    elif name == "dc":
        x = np.full(n, 0.5)
    elif name == "sine_bin":          # full scale, exactly on bin 100 of a 2048-point frame
        x = np.sin(2 * np.pi * 100.0 * t / W)
    elif name == "sine_half":         # halfway between bins 100 and 101
        x = np.sin(2 * np.pi * 100.5 * t / W)
    elif name == "two_tones":         # 100 dB apart
        x = np.sin(2 * np.pi * 100.25 * t / W) + 1e-5 * np.sin(2 * np.pi * 700.5 * t / W + 0.3)
    elif name == "impulse":           # unit impulses on, and one sample before, hop boundaries
        x = np.zeros(n)
        for p in (HOP, 2 * HOP, 3 * HOP - 1, n - 1):
            if 0 <= p < n:
                x[p] = 1.0
    elif name == "chirp":             # linear sweep 0 .. fs/2 over the clip
        x = 0.8 * np.sin(np.pi * (t * t) / (2.0 * n))
    elif name == "noise_m90":         # white noise at -90 dBFS
        x = np.random.default_rng([90, n]).standard_normal(n) * 10.0 ** (-90.0 / 20.0)
    elif name == "clipped_pcm":
        x = clipped_pcm16(n).astype(np.float64) / 32768.0
    else:
        raise KeyError(name)
    return x.astype(np.float32)
