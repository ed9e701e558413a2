"""CPU oracle for the zaf.py windowed-transform hot path.  TEST INFRASTRUCTURE ONLY.

This module is a float64 NumPy/SciPy restatement of the reference algorithm
(zafarrafii/Zaf-Python, `zaf.py`).  It exists so that the HIP kernels can be
checked on a machine where the reference itself is absent (the GPU box).

Rules (see DESIGN.md "Oracle"):
  * only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
    leg may import it; the product (`zaf-python_amd/zafx`) never does;
  * parity is PINNED: `tests/golden/make_golden.py` imports the real
    `/root/reference/zaf.py` in the build container and commits golden vectors;
    `tests/test_oracle_golden.py` checks every function below against them
    (bit-identical on NumPy 2.2.6 / SciPy 1.15.3, <= 1e-12 normwise elsewhere).

Every function cites the reference lines it restates.  The arithmetic is the
same sequence of NumPy calls (so results are bit-identical on the same NumPy),
but framing is written with explicit index arithmetic instead of the
reference's running-offset Python loops.
"""

import numpy as np
import scipy.fftpack
import scipy.sparse

__all__ = [
    "stft", "istft", "melfilterbank", "melspectrogram", "mfcc",
    "cqtkernel", "cqtspectrogram", "cqtchromagram", "mdct", "imdct",
    "stft_num_frames", "mdct_num_frames", "hamming_periodic", "kbd_window",
    "sine_window", "stft_batch", "mdct_batch",
]


# --------------------------------------------------------------------------
# windows used by the BASELINE configs (not part of zaf.py)
# --------------------------------------------------------------------------
def hamming_periodic(window_length):
    """Periodic Hamming, == scipy.signal.windows.hamming(W, sym=False)."""
    n = np.arange(window_length)
    return 0.54 - 0.46 * np.cos(2.0 * np.pi * n / window_length)


def kbd_window(window_length, alpha=5.0):
    """Kaiser-Bessel-derived window (Princen-Bradley compliant), beta = alpha*pi.

    Equivalent to scipy.signal.windows.kaiser_bessel_derived(W, beta=alpha*pi);
    the off-by-one docstring recipe of zaf.py:1004-1010 is NOT reproduced
    (SURVEY 8a "quirks").
    """
    half = window_length // 2
    kaiser = np.kaiser(half + 1, alpha * np.pi)
    csum = np.cumsum(kaiser[:half])
    half_win = np.sqrt(csum / np.sum(kaiser))
    return np.concatenate((half_win, half_win[::-1]))


def sine_window(window_length):
    """Sine window (zaf.py:1098-1109 uses a sine-power/Vorbis window; plain sine is PB too)."""
    return np.sin(np.pi / window_length * (np.arange(window_length) + 0.5))


# --------------------------------------------------------------------------
# frame-count formulas (integer arithmetic that must be bit-exact)
# --------------------------------------------------------------------------
def stft_num_frames(number_samples, window_length, step_length):
    """zaf.py:99-109 : T = ceil((N + 2*floor(W/2) - W)/H) + 1."""
    padding_length = int(np.floor(window_length / 2))
    return int(np.ceil((number_samples + 2 * padding_length - window_length) / step_length)) + 1


def mdct_num_frames(number_samples, window_length):
    """zaf.py:1029-1033 : T = ceil(N/(W/2)) + 1."""
    return int(np.ceil(number_samples / int(window_length / 2))) + 1


# --------------------------------------------------------------------------
# a1  stft  (zaf.py:45-141)
# --------------------------------------------------------------------------
def stft(audio_signal, window_function, step_length):
    """Restates zaf.py:95-141.  Returns (W, T) complex128, two-sided."""
    x = np.asarray(audio_signal)
    w = np.asarray(window_function)
    n = len(x)
    wl = len(w)
    pad = int(np.floor(wl / 2))                                   # :99
    nt = stft_num_frames(n, wl, step_length)                      # :102-109
    total = nt * step_length + (wl - step_length)                 # :117-121
    xp = np.zeros(total, dtype=np.result_type(x.dtype, np.float64))
    xp[pad:pad + n] = x                                           # :112-125
    frames = np.zeros((wl, nt))                                   # :128
    for j in range(nt):                                           # :132-136
        frames[:, j] = xp[j * step_length:j * step_length + wl] * w
    return np.fft.fft(frames, axis=0)                             # :139


def stft_batch(clips, window_function, step_length):
    """Vectorised (B, N) -> (B, W, T) form of `stft` (same arithmetic per clip)."""
    clips = np.atleast_2d(np.asarray(clips, dtype=np.float64))
    w = np.asarray(window_function, dtype=np.float64)
    b, n = clips.shape
    wl = len(w)
    pad = wl // 2
    nt = stft_num_frames(n, wl, step_length)
    total = nt * step_length + (wl - step_length)
    xp = np.zeros((b, total))
    xp[:, pad:pad + n] = clips
    idx = (np.arange(nt) * step_length)[:, None] + np.arange(wl)[None, :]
    frames = xp[:, idx] * w                                       # (B, T, W)
    return np.fft.fft(frames, axis=-1).transpose(0, 2, 1)


# --------------------------------------------------------------------------
# a2  istft  (zaf.py:144-243)
# --------------------------------------------------------------------------
def istft(audio_stft, window_function, step_length):
    """Restates zaf.py:214-243.  Returns (T*H - (W-H),) float64."""
    spec = np.asarray(audio_stft)
    w = np.asarray(window_function)
    wl, nt = spec.shape                                           # :214
    total = nt * step_length + (wl - step_length)                 # :217
    y = np.zeros(total)                                           # :220
    frames = np.real(np.fft.ifft(spec, axis=0))                   # :223
    for j in range(nt):                                           # :226-233
        lo = j * step_length
        y[lo:lo + wl] = y[lo:lo + wl] + frames[:, j]
    y = y[wl - step_length:total - (wl - step_length)]            # :236-238
    return y / sum(w[0:wl:step_length])                           # :241


# --------------------------------------------------------------------------
# a3  melfilterbank  (zaf.py:246-321)
# --------------------------------------------------------------------------
def melfilterbank(sampling_frequency, window_length, number_filters):
    """Restates zaf.py:280-321.  Returns CSR float64 (n_filters, W/2)."""
    mel_lo = 2595 * np.log10(1 + (sampling_frequency / window_length) / 700)   # :280
    mel_hi = 2595 * np.log10(1 + (sampling_frequency / 2) / 700)               # :281
    width = 2 * (mel_hi - mel_lo) / (number_filters + 1)                       # :284
    mel_pts = np.arange(mel_lo, mel_hi + 1, width / 2)                         # :287
    edges = np.round(
        700 * (np.power(10, mel_pts / 2595) - 1) * window_length / sampling_frequency
    ).astype(int)                                                              # :290-295
    fb = np.zeros((number_filters, int(window_length / 2)))                    # :298
    for i in range(number_filters):                                            # :301-316
        a, b, c = edges[i], edges[i + 1], edges[i + 2]
        fb[i, a - 1:b] = np.linspace(0, 1, num=b - a + 1)
        fb[i, b - 1:c] = np.linspace(1, 0, num=c - b + 1)
    return scipy.sparse.csr_matrix(fb)                                         # :319


# --------------------------------------------------------------------------
# a4  melspectrogram  (zaf.py:324-375)
# --------------------------------------------------------------------------
def melspectrogram(audio_signal, window_function, step_length, mel_filterbank):
    """Restates zaf.py:369-375.  Returns (n_mels, T) float64."""
    spec = stft(audio_signal, window_function, step_length)                    # :369
    mag = abs(spec[1:int(len(window_function) / 2) + 1, :])                    # :370
    return np.matmul(mel_filterbank.toarray(), mag)                            # :373


# --------------------------------------------------------------------------
# a5  mfcc  (zaf.py:378-454)
# --------------------------------------------------------------------------
def mfcc(audio_signal, window_function, step_length, mel_filterbank, number_coefficients):
    """Restates zaf.py:436-454.  Returns (ncoef, T) float64."""
    spec = stft(audio_signal, window_function, step_length)                    # :436
    power = np.power(abs(spec[1:int(len(window_function) / 2) + 1, :]), 2)     # :437-439
    logmel = np.log(np.matmul(mel_filterbank.toarray(), power) + np.finfo(float).eps)  # :444-446
    coefs = scipy.fftpack.dct(logmel, axis=0, norm="ortho")                    # :443-449
    return coefs[1:number_coefficients + 1, :]                                 # :452


# --------------------------------------------------------------------------
# a6  cqtkernel  (zaf.py:457-559)
# --------------------------------------------------------------------------
def cqtkernel(sampling_frequency, octave_resolution, minimum_frequency, maximum_frequency):
    """Restates zaf.py:497-559.  Returns CSR complex128 (n_bins, fft_len)."""
    q = 1 / (pow(2, 1 / octave_resolution) - 1)                                # :497
    nbins = round(octave_resolution * np.log2(maximum_frequency / minimum_frequency))  # :500-502
    fft_len = int(pow(2, np.ceil(np.log2(q * sampling_frequency / minimum_frequency))))  # :505-509
    kern = np.zeros((nbins, fft_len), dtype=complex)                           # :512
    for i in range(nbins):                                                     # :515-544
        freq = minimum_frequency * pow(2, i / octave_resolution)               # :518
        wl = 2 * round(q * sampling_frequency / freq / 2) + 1                  # :521-523
        t = np.arange(-(wl - 1) / 2, (wl - 1) / 2 + 1)
        atom = np.hamming(wl) * np.exp(2 * np.pi * 1j * q * t / wl) / wl       # :526-537
        lo = int((fft_len - wl + 1) / 2)                                       # :540
        kern[i, lo:lo + wl] = atom                                             # :544
    kern = np.fft.fft(kern, axis=1)                                            # :548
    kern[np.absolute(kern) < 0.01] = 0                                         # :551
    kern = scipy.sparse.csr_matrix(kern)                                       # :554
    return np.conjugate(kern) / fft_len                                        # :557


# --------------------------------------------------------------------------
# a7  cqtspectrogram  (zaf.py:562-635)
# --------------------------------------------------------------------------
def cqtspectrogram(audio_signal, sampling_frequency, time_resolution, cqt_kernel):
    """Restates zaf.py:603-635.  Returns (n_bins, T) float64."""
    x = np.asarray(audio_signal)
    step = round(sampling_frequency / time_resolution)                         # :603
    nt = int(np.floor(len(x) / step))                                          # :606
    nbins, fft_len = np.shape(cqt_kernel)                                      # :609
    left = int(np.ceil((fft_len - step) / 2))                                  # :615
    right = int(np.floor((fft_len - step) / 2))                                # :616
    xp = np.zeros(left + len(x) + right, dtype=np.result_type(x.dtype, np.float64))
    xp[left:left + len(x)] = x                                                 # :612-620
    out = np.zeros((nbins, nt))                                                # :623
    for j in range(nt):                                                        # :627-633
        seg = xp[j * step:j * step + fft_len]
        out[:, j] = np.absolute(cqt_kernel * np.fft.fft(seg))                  # :630-632 (CSR mat-vec)
    return out


# --------------------------------------------------------------------------
# 8(f) rank 1  cqtchromagram  (zaf.py:638-700)
# --------------------------------------------------------------------------
def cqtchromagram(audio_signal, sampling_frequency, time_resolution, octave_resolution, cqt_kernel):
    """Restates zaf.py:682-700.  Returns (octave_resolution, T) float64."""
    spec = cqtspectrogram(audio_signal, sampling_frequency, time_resolution, cqt_kernel)  # :682
    nbins, nt = np.shape(spec)
    chroma = np.zeros((octave_resolution, nt))                                 # :690
    for i in range(octave_resolution):                                         # :693-698
        chroma[i, :] = np.sum(spec[i:nbins:octave_resolution, :], axis=0)
    return chroma


# --------------------------------------------------------------------------
# a8  mdct  (zaf.py:984-1075)
# --------------------------------------------------------------------------
def _mdct_twiddles(window_length):
    pre = np.exp(-1j * np.pi / window_length * np.arange(0, window_length))    # :1047-1049
    post = np.exp(
        -1j * np.pi / window_length * (window_length / 2 + 1)
        * np.arange(0.5, window_length / 2 + 0.5)
    )                                                                          # :1050-1056
    return pre, post


def mdct(audio_signal, window_function):
    """Restates zaf.py:1025-1075.  Returns (W/2, T) float64."""
    x = np.asarray(audio_signal)
    w = np.asarray(window_function)
    n = len(x)
    wl = len(w)
    hop = int(wl / 2)                                                          # :1029
    nf = int(wl / 2)                                                           # :1030
    nt = mdct_num_frames(n, wl)                                                # :1033
    xp = np.zeros(hop + n + ((nt + 1) * hop - n), dtype=np.result_type(x.dtype, np.float64))
    xp[hop:hop + n] = x                                                        # :1036-1041
    out = np.zeros((nf, nt))                                                   # :1044
    pre, post = _mdct_twiddles(wl)
    for j in range(nt):                                                        # :1061-1073
        seg = xp[j * hop:j * hop + wl] * w                                     # :1064
        seg = np.fft.fft(seg * pre)                                            # :1068
        out[:, j] = np.real(seg[0:nf] * post)                                  # :1071-1073
    return out


def mdct_batch(clips, window_function):
    """Vectorised (B, N) -> (B, W/2, T) form of `mdct`."""
    clips = np.atleast_2d(np.asarray(clips, dtype=np.float64))
    w = np.asarray(window_function, dtype=np.float64)
    b, n = clips.shape
    wl = len(w)
    hop = wl // 2
    nt = mdct_num_frames(n, wl)
    xp = np.zeros((b, (nt + 2) * hop))
    xp[:, hop:hop + n] = clips
    idx = (np.arange(nt) * hop)[:, None] + np.arange(wl)[None, :]
    pre, post = _mdct_twiddles(wl)
    seg = np.fft.fft(xp[:, idx] * w * pre, axis=-1)
    return np.real(seg[..., :hop] * post).transpose(0, 2, 1)


# --------------------------------------------------------------------------
# a9  imdct  (zaf.py:1078-1184)
# --------------------------------------------------------------------------
def imdct(audio_mdct, window_function):
    """Restates zaf.py:1125-1184.  Returns (H*(T-1) - 1,) float64."""
    coefs = np.asarray(audio_mdct)
    w = np.asarray(window_function)
    nf, nt = coefs.shape                                                       # :1125
    wl = 2 * nf                                                                # :1128
    hop = nf                                                                   # :1129
    total = hop * (nt + 1)                                                     # :1132
    y = np.zeros(total)                                                        # :1135
    pre = np.exp(-1j * np.pi / (2 * nf) * (nf + 1) * np.arange(0, nf))         # :1138-1144
    post = np.exp(
        -1j * np.pi / (2 * nf) * np.arange(0.5 + nf / 2, 2 * nf + nf / 2 + 0.5)
    ) / nf                                                                     # :1145-1156
    spec = np.fft.fft(coefs * pre[:, np.newaxis], n=2 * nf, axis=0)            # :1159-1163
    frames = 2 * (np.real(spec * post[:, np.newaxis]) * w[:, np.newaxis])      # :1166-1169
    for j in range(nt):                                                        # :1173-1179
        lo = j * hop
        y[lo:lo + wl] = y[lo:lo + wl] + frames[:, j]
    return y[hop:-hop - 1]                                                     # :1182


# --------------------------------------------------------------------------
# 8(f) rank 3  dct / dst, types I-IV, orthonormal  (zaf.py:703-981)
# --------------------------------------------------------------------------
def _ext_fft(parts, length):
    """Scatter (offset, stride, values) pieces into a zero vector of `length` and FFT it."""
    ext = np.zeros(length)
    for start, stop, step, values in parts:
        ext[start:stop:step] = values
    return np.fft.fft(ext)


def dct(audio_signal, dct_type):
    """Restates zaf.py:759-839: DCT-I..IV by FFT of a symmetric extension, scaled to be orthonormal."""
    x = np.array(audio_signal, dtype=float)
    n = len(x)
    if dct_type == 1:
        x[[0, -1]] = x[[0, -1]] * np.sqrt(2)                                   # :766-767
        spec = np.fft.fft(np.concatenate((x, x[-2:0:-1])))                     # :770-771
        out = np.real(spec[0:n]) / 2                                           # :772
        out[[0, -1]] = out[[0, -1]] / np.sqrt(2)                               # :775
        return out * np.sqrt(2 / (n - 1))                                      # :776
    if dct_type == 2:
        spec = _ext_fft([(1, 2 * n, 2, x), (2 * n + 1, 4 * n, 2, x[::-1])], 4 * n)   # :786-789
        out = np.real(spec[0:n]) / 2                                           # :790
        out[0] = out[0] / np.sqrt(2)                                           # :793
        return out * np.sqrt(2 / n)                                            # :794
    if dct_type == 3:
        x[0] = x[0] * np.sqrt(2)                                               # :806
        spec = _ext_fft([(0, n, 1, x), (n + 1, 2 * n + 1, 1, -x[::-1]), (2 * n + 1, 3 * n, 1, -x[1:]),
                         (3 * n + 1, 4 * n, 1, x[:0:-1])], 4 * n)              # :809-814
        return np.real(spec[1:2 * n:2]) / 4 * np.sqrt(2 / n)                   # :815-818
    if dct_type == 4:
        spec = _ext_fft([(1, 2 * n, 2, x), (2 * n + 1, 4 * n, 2, -x[::-1]), (4 * n + 1, 6 * n, 2, -x),
                         (6 * n + 1, 8 * n, 2, x[::-1])], 8 * n)               # :828-833
        return np.real(spec[1:2 * n:2]) / 4 * np.sqrt(2 / n)                   # :834-837
    raise ValueError("dct_type must be 1, 2, 3 or 4")


def dst(audio_signal, dst_type):
    """Restates zaf.py:901-981: DST-I..IV by FFT of an antisymmetric extension, orthonormal."""
    x = np.array(audio_signal, dtype=float)
    n = len(x)
    if dst_type == 1:
        spec = _ext_fft([(1, n + 1, 1, x), (n + 2, 2 * n + 2, 1, -x[::-1])], 2 * n + 2)   # :907-910
        return -np.imag(spec[1:n + 1]) / 2 * np.sqrt(2 / (n + 1))              # :911-914
    if dst_type == 2:
        spec = _ext_fft([(1, 2 * n, 2, x), (2 * n + 1, 4 * n, 2, -x[::-1])], 4 * n)       # :924-927
        out = -np.imag(spec[1:n + 1]) / 2                                      # :928
        out[-1] = out[-1] / np.sqrt(2)                                         # :931
        return out * np.sqrt(2 / n)                                            # :932
    if dst_type == 3:
        x[-1] = x[-1] * np.sqrt(2)                                             # :944
        spec = _ext_fft([(1, n + 1, 1, x), (n + 1, 2 * n, 1, x[-2::-1]), (2 * n + 1, 3 * n + 1, 1, -x),
                         (3 * n + 1, 4 * n, 1, -x[-2::-1])], 4 * n)            # :947-952
        return -np.imag(spec[1:2 * n:2]) / 4 * np.sqrt(2 / n)                  # :953-956
    if dst_type == 4:
        spec = _ext_fft([(1, 2 * n, 2, x), (2 * n + 1, 4 * n, 2, x[::-1]), (4 * n + 1, 6 * n, 2, -x),
                         (6 * n + 1, 8 * n, 2, -x[::-1])], 8 * n)              # :964-975
        return -np.imag(spec[1:2 * n:2]) / 4 * np.sqrt(2 / n)                  # :976-979
    raise ValueError("dst_type must be 1, 2, 3 or 4")
