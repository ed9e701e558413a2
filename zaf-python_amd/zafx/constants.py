"""Host-side constant builders of the windowed-transform path (NumPy, float64).

These are one-time, microsecond-to-second host computations whose RESULTS are the
operands of the device kernels: the mel filterbank, the CQT
spectral kernel, the DCT-II rows used by `mfcc`, and the analysis windows of the
BASELINE configs.  Signatures and return types follow zaf.py (scipy CSR matrices).
"""
import numpy as np
import scipy.sparse

__all__ = ["melfilterbank", "cqtkernel", "dct2_rows", "dct_matrix", "dst_matrix", "hamming", "kaiser_bessel_derived", "sine"]


def melfilterbank(sampling_frequency, window_length, number_filters):
    """Triangular mel filterbank, CSR float64 (number_filters, window_length/2).

    Same contract as zaf.melfilterbank (zaf.py:246-321): filter edges are equally
    spaced on the mel scale 2595*log10(1 + f/700) between fs/W and fs/2, rounded to
    1-based FFT-bin numbers; filter i rises linearly from edge i to edge i+1 and
    falls to edge i+2 (peak 1.0, rows not area-normalised); column c is FFT bin c+1.
    """
    def hz_to_mel(f):
        return 2595 * np.log10(1 + f / 700)

    lowest, highest = hz_to_mel(sampling_frequency / window_length), hz_to_mel(sampling_frequency / 2)
    half_width = (highest - lowest) / (number_filters + 1)
    mel_edges = np.arange(lowest, highest + 1, half_width)
    bins = np.round(700 * (np.power(10, mel_edges / 2595) - 1) * window_length / sampling_frequency).astype(int)
    bank = np.zeros((number_filters, window_length // 2))
    for row in range(number_filters):
        left, peak, right = bins[row:row + 3]
        bank[row, left - 1:peak] = np.linspace(0, 1, peak - left + 1)
        bank[row, peak - 1:right] = np.linspace(1, 0, right - peak + 1)
    return scipy.sparse.csr_matrix(bank)


def cqtkernel(sampling_frequency, octave_resolution, minimum_frequency, maximum_frequency):
    """Sparse spectral CQT kernel, CSR complex128 (n_bins, fft_length).

    Same contract as zaf.cqtkernel (zaf.py:457-559): Q = 1/(2^(1/r) - 1); for every
    bin a Hamming-windowed complex exponential of odd length 2*round(Q fs/f/2)+1,
    centred in a frame of fft_length = 2^ceil(log2(Q fs/fmin)); FFT along the frame;
    magnitudes < 0.01 zeroed; result conj(.)/fft_length.
    """
    quality = 1 / (pow(2, 1 / octave_resolution) - 1)
    n_bins = round(octave_resolution * np.log2(maximum_frequency / minimum_frequency))
    fft_length = int(pow(2, np.ceil(np.log2(quality * sampling_frequency / minimum_frequency))))
    atoms = np.zeros((n_bins, fft_length), dtype=complex)
    for b in range(n_bins):
        centre_hz = minimum_frequency * pow(2, b / octave_resolution)
        length = 2 * round(quality * sampling_frequency / centre_hz / 2) + 1
        half = (length - 1) / 2
        phase = 2 * np.pi * 1j * quality * np.arange(-half, half + 1) / length
        start = int((fft_length - length + 1) / 2)
        atoms[b, start:start + length] = np.hamming(length) * np.exp(phase) / length
    spectra = np.fft.fft(atoms, axis=1)
    spectra[np.absolute(spectra) < 0.01] = 0
    return np.conjugate(scipy.sparse.csr_matrix(spectra)) / fft_length


def dct2_rows(number_filters, number_coefficients):
    """Rows 1..number_coefficients of the orthonormal DCT-II matrix of size number_filters.

    scipy.fftpack.dct(y, axis=0, norm="ortho")[1:ncoef+1] == dct2_rows(M, ncoef) @ y
    (zaf.py:443-452): C[q, m] = sqrt(2/M) cos(pi q (2m+1) / (2M)), q >= 1.
    """
    q = np.arange(1, number_coefficients + 1)[:, None]
    m = np.arange(number_filters)[None, :]
    return np.sqrt(2.0 / number_filters) * np.cos(np.pi * q * (2 * m + 1) / (2 * number_filters))


def hamming(window_length, periodic=True):
    """Hamming window; periodic=True equals scipy.signal.windows.hamming(W, sym=False)."""
    denom = window_length if periodic else window_length - 1
    return 0.54 - 0.46 * np.cos(2 * np.pi * np.arange(window_length) / denom)


def kaiser_bessel_derived(window_length, alpha=5.0):
    """Princen-Bradley compliant KBD window (beta = alpha*pi), as scipy's kaiser_bessel_derived."""
    half = window_length // 2
    kaiser = np.kaiser(half + 1, alpha * np.pi)
    rising = np.sqrt(np.cumsum(kaiser[:half]) / np.sum(kaiser))
    return np.concatenate((rising, rising[::-1]))


def sine(window_length):
    """Sine window (Princen-Bradley compliant)."""
    return np.sin(np.pi / window_length * (np.arange(window_length) + 0.5))


def dct_matrix(length, dct_type):
    """Orthonormal DCT matrix of type 1-4: zaf.dct(x, t) == dct_matrix(len(x), t) @ x (zaf.py:703-839).

    I:   sqrt(2/(N-1)) a_k a_n cos(pi k n / (N-1)),  a_0 = a_{N-1} = 1/sqrt(2)
    II:  sqrt(2/N) c_k cos(pi k (2n+1) / (2N)),       c_0 = 1/sqrt(2)
    III: transpose of II
    IV:  sqrt(2/N) cos(pi (2k+1)(2n+1) / (4N))
    """
    n = int(length)
    k = np.arange(n)[:, None]
    m = np.arange(n)[None, :]
    if dct_type == 1:
        if n < 2:
            raise ValueError("DCT-I needs at least 2 samples")
        edge = np.ones(n)
        edge[[0, -1]] = 1 / np.sqrt(2)
        return np.sqrt(2 / (n - 1)) * edge[:, None] * edge[None, :] * np.cos(np.pi * k * m / (n - 1))
    if dct_type in (2, 3):
        first = np.ones(n)
        first[0] = 1 / np.sqrt(2)
        mat = np.sqrt(2 / n) * first[:, None] * np.cos(np.pi * k * (2 * m + 1) / (2 * n))
        return mat if dct_type == 2 else mat.T
    if dct_type == 4:
        return np.sqrt(2 / n) * np.cos(np.pi * (2 * k + 1) * (2 * m + 1) / (4 * n))
    raise ValueError("dct_type must be 1, 2, 3 or 4")


def dst_matrix(length, dst_type):
    """Orthonormal DST matrix of type 1-4: zaf.dst(x, t) == dst_matrix(len(x), t) @ x (zaf.py:842-981).

    I:   sqrt(2/(N+1)) sin(pi (k+1)(n+1) / (N+1))
    II:  sqrt(2/N) c_k sin(pi (k+1)(2n+1) / (2N)),    c_{N-1} = 1/sqrt(2)
    III: transpose of II
    IV:  sqrt(2/N) sin(pi (2k+1)(2n+1) / (4N))
    """
    n = int(length)
    k = np.arange(n)[:, None]
    m = np.arange(n)[None, :]
    if dst_type == 1:
        return np.sqrt(2 / (n + 1)) * np.sin(np.pi * (k + 1) * (m + 1) / (n + 1))
    if dst_type in (2, 3):
        last = np.ones(n)
        last[-1] = 1 / np.sqrt(2)
        mat = np.sqrt(2 / n) * last[:, None] * np.sin(np.pi * (k + 1) * (2 * m + 1) / (2 * n))
        return mat if dst_type == 2 else mat.T
    if dst_type == 4:
        return np.sqrt(2 / n) * np.sin(np.pi * (2 * k + 1) * (2 * m + 1) / (4 * n))
    raise ValueError("dst_type must be 1, 2, 3 or 4")
