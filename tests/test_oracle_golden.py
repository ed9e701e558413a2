"""Pin the CPU oracle (oracle/zaf_oracle.py) to golden vectors produced by the real
reference (tests/golden/make_golden.py).  CPU-only; runs everywhere."""
import numpy as np
import pytest
import scipy.sparse

from conftest import excess, mfcc_floor, relerr, synth_clip
from oracle import zaf_oracle as orc

TOL = 1e-12  # normwise; bit-identical on the NumPy the fixtures were made with


def close(a, b, tol=TOL):
    assert relerr(a, b) <= tol, relerr(a, b)


def csr(g, tag):
    return scipy.sparse.csr_matrix(
        (g[f"{tag}_data"], g[f"{tag}_indices"], g[f"{tag}_indptr"]), shape=tuple(g[f"{tag}_shape"])
    )


def test_windows_match_scipy():
    import scipy.signal.windows as sw
    assert np.allclose(orc.hamming_periodic(2048), sw.hamming(2048, sym=False), atol=1e-15)
    assert np.allclose(orc.kbd_window(2048), sw.kaiser_bessel_derived(2048, beta=5 * np.pi), atol=1e-14)
    w = orc.kbd_window(2048)
    assert np.max(np.abs(w[:1024] ** 2 + w[1024:] ** 2 - 1)) < 1e-13  # Princen-Bradley
    s = orc.sine_window(64)
    assert np.max(np.abs(s[:32] ** 2 + s[32:] ** 2 - 1)) < 1e-13


def test_constants(golden):
    g = golden["consts"]
    for tag, args in (("fb128", (44100, 2048, 128)), ("fb40", (44100, 2048, 40))):
        ref = csr(g, tag)
        got = orc.melfilterbank(*args)
        assert got.shape == ref.shape and got.nnz == ref.nnz
        close(got.toarray(), ref.toarray())
    ref = csr(g, "ck_small")
    got = orc.cqtkernel(4000, 12, 200, 1600)
    assert got.shape == ref.shape and got.nnz == ref.nnz
    close(got.toarray(), ref.toarray())


@pytest.mark.timeout(120)
def test_cqtkernel_config(golden):
    g = golden["consts"]
    ref = csr(g, "ck")
    got = orc.cqtkernel(44100, 24, 55, 3520).tocsr()
    got.sort_indices()
    assert got.shape == (144, 32768) and got.nnz == ref.nnz == 9450
    assert np.array_equal(got.indices, ref.indices) and np.array_equal(got.indptr, ref.indptr)
    close(got.data, ref.data)


@pytest.mark.timeout(180)
def test_cqtkernel_docstring_example(golden):
    """The kernel of zaf.py:476-483 (55 Hz ... fs/2): columns on both halves of the spectrum, 60 879 non-zeros."""
    g = golden["cqtfull"]
    got = orc.cqtkernel(44100, 24, 55, 44100 / 2).tocsr()
    got.sort_indices()
    assert got.shape == tuple(g["shape"]) == (208, 32768)
    assert np.array_equal(got.indptr, g["indptr"]) and got.nnz == 60879
    assert got.indices.min() == g["col_min"] == 39 and got.indices.max() == g["col_max"] == 16613
    assert np.array_equal(got.indices[g["probe_idx"]], g["probe_col"])
    close(got.data[g["probe_idx"]], g["probe_val"])
    x = synth_clip(5, 0, 100000).astype(np.float64)
    close(orc.cqtspectrogram(x, 44100, 25, got), g["cqt"])
    close(orc.cqtchromagram(x, 44100, 25, 24, got), g["chroma"])


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000])
@pytest.mark.parametrize("hop", [32, 16])
def test_tiny_stft_family(golden, n, hop):
    g = golden["tiny"]
    x, ham = g[f"x_{n}"], g["ham"]
    fb = scipy.sparse.csr_matrix(g["fb_dense"])
    s = orc.stft(x, ham, hop)
    close(s, g[f"stft_{n}_{hop}"])
    close(orc.stft_batch(x[None], ham, hop)[0], g[f"stft_{n}_{hop}"])
    close(orc.istft(s, ham, hop), g[f"istft_{n}_{hop}"])
    close(orc.melspectrogram(x, ham, hop, fb), g[f"mel_{n}_{hop}"])
    close(orc.mfcc(x, ham, hop, fb, 5), g[f"mfcc_{n}_{hop}"], 1e-11)


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000])
@pytest.mark.parametrize("wname", ["sine", "kbd"])
def test_tiny_mdct_family(golden, n, wname):
    g = golden["tiny"]
    x, w = g[f"x_{n}"], g[wname]
    m = orc.mdct(x, w)
    close(m, g[f"mdct_{wname}_{n}"])
    close(orc.mdct_batch(x[None], w)[0], g[f"mdct_{wname}_{n}"])
    y = orc.imdct(m, w)
    close(y, g[f"imdct_{wname}_{n}"])
    # TDAC perfect reconstruction (zaf.py:1098-1109); output is one sample short (zaf.py:1182)
    k = min(n, len(y))
    assert np.max(np.abs(y[:k] - x[:k])) < 1e-12


def test_tiny_istft_generic(golden):
    g = golden["tiny"]
    close(orc.istft(g["istft_generic_in"], g["ham"], 32), g["istft_generic_out"])


@pytest.mark.parametrize("n", [400, 4000, 4321])
def test_tiny_cqt(golden, n):
    g = golden["tiny"]
    ck = scipy.sparse.csr_matrix(g["ck_dense"])
    x = g[f"xq_{n}"]
    close(orc.cqtspectrogram(x, 4000, 50, ck), g[f"cqt_{n}"])
    close(orc.cqtchromagram(x, 4000, 50, 12, ck), g[f"chroma_{n}"])


def test_lengths_table(golden):
    tab = golden["lengths"]["table"]
    for n, t_stft, t_mdct, l_istft, l_imdct in tab:
        assert orc.stft_num_frames(int(n), 2048, 1024) == t_stft
        assert orc.mdct_num_frames(int(n), 2048) == t_mdct
        assert t_stft * 1024 - 1024 == l_istft
        assert 1024 * (t_mdct - 1) - 1 == l_imdct
    # SURVEY section 4 normative rows
    rows = {int(r[0]): tuple(int(v) for v in r[1:]) for r in tab}
    assert rows[1] == (2, 2, 1024, 1023)
    assert rows[2048] == (3, 3, 2048, 2047)
    assert rows[2049] == (4, 4, 3072, 3071)
    assert rows[441000] == (432, 432, 441344, 441343)


def _check_probes(g, key, arr, tol=TOL):
    assert tuple(g[f"{key}_shape"]) == arr.shape
    flat = arr.reshape(-1)
    scale = float(g[f"{key}_maxabs"])
    assert np.max(np.abs(flat[g[f"{key}_idx"]] - g[f"{key}_val"])) <= tol * scale
    if arr.ndim == 2:
        assert np.max(np.abs(arr.sum(axis=1) - g[f"{key}_rowsum"])) <= 1e-9 * scale * arr.shape[1]
        assert np.max(np.abs(arr.sum(axis=0) - g[f"{key}_colsum"])) <= 1e-9 * scale * arr.shape[0]
    assert abs(np.sqrt(np.sum(np.abs(arr) ** 2)) - float(g[f"{key}_l2"])) <= 1e-10 * float(g[f"{key}_l2"])


@pytest.mark.timeout(300)
@pytest.mark.parametrize("c", [0, 1])
def test_config_S(golden, c):
    """BASELINE configs 1-4 at full clip size (10 s @ 44.1 kHz, W 2048, hop 1024)."""
    g = golden["config"]
    x = synth_clip(0, c, 441000).astype(np.float64)
    assert x.sum() == float(g[f"S{c}_x_sum"]) and np.array_equal(x[:16], g[f"S{c}_x_head"]), \
        "NumPy Generator stream changed: regenerate tests/golden"
    ham = orc.hamming_periodic(2048)
    kbd = orc.kbd_window(2048)
    fb = orc.melfilterbank(44100, 2048, 128)
    s = orc.stft(x, ham, 1024)
    _check_probes(g, f"S{c}_stft", s)
    _check_probes(g, f"S{c}_istft", orc.istft(s, ham, 1024))
    _check_probes(g, f"S{c}_mel", orc.melspectrogram(x, ham, 1024, fb), 1e-11)
    _check_probes(g, f"S{c}_mfcc", orc.mfcc(x, ham, 1024, fb, 20), 1e-10)
    m = orc.mdct(x, kbd)
    _check_probes(g, f"S{c}_mdct", m, 1e-11)
    y = orc.imdct(m, kbd)
    _check_probes(g, f"S{c}_imdct", y, 1e-11)
    assert np.max(np.abs(y[:440999] - x[:440999])) < 1e-11  # round trip, SURVEY section 4
    # Hermitian symmetry of the two-sided spectrum
    assert np.max(np.abs(s[1:1024] - np.conj(s[:1024:-1]))) < 1e-11
    assert np.all(s[0].imag == 0) and np.all(s[1024].imag == 0)


@pytest.mark.timeout(300)
def test_config_Q(golden):
    """BASELINE config 5 clip: 30 s, 24 bins/octave, 55-3520 Hz, 25 frames/s."""
    g = golden["config"]
    x = synth_clip(0, 0, 1323000).astype(np.float64)
    assert x.sum() == float(g["Q0_x_sum"])
    ck = csr(golden["consts"], "ck")
    q = orc.cqtspectrogram(x, 44100, 25, ck)
    _check_probes(g, "Q0_cqt", q, 1e-11)
    _check_probes(g, "Q0_chroma", orc.cqtchromagram(x, 44100, 25, 24, ck), 1e-11)


@pytest.mark.parametrize("n", [8, 9, 100, 1024, 63, 64, 65, 1023, 1025])
def test_dct_dst(golden, n):
    """oracle dct/dst types 1-4 against the reference, plus the reference's own
    plotted self-checks (zaf.py:728-753 DCT vs SciPy ortho; :866-897 DST inverse pairs)."""
    import scipy.fftpack
    g = golden["dctdst"]
    x = g[f"x_{n}"]
    for t in (1, 2, 3, 4):
        close(orc.dct(x, t), g[f"dct{t}_{n}"])
        close(orc.dst(x, t), g[f"dst{t}_{n}"])
        close(orc.dct(x, t), scipy.fftpack.dct(x, type=t, norm="ortho"), 1e-11)
    close(orc.dst(orc.dst(x, 1), 1), x, 1e-11)
    close(orc.dst(orc.dst(x, 2), 3), x, 1e-11)
    close(orc.dst(orc.dst(x, 4), 4), x, 1e-11)


@pytest.mark.parametrize("name", __import__("signals").NAMES)
def test_signals_that_are_not_noise(golden, name):
    """tests/signals.py through the oracle against the real reference's outputs (signals.npz): silence, DC, tones, impulses,
    a chirp, -90 dBFS noise, clipped PCM.  Also pins that the recipe still produces the input the fixture was made from."""
    import signals as sig
    g = golden["signals"]
    x = sig.signal(name, sig.N_FRAMES).astype(np.float64)
    assert x.sum() == g[f"{name}_x_sum"] and np.abs(x).sum() == g[f"{name}_x_abs"]
    ham, kbd = orc.hamming_periodic(sig.W), orc.kbd_window(sig.W)
    fb = orc.melfilterbank(sig.FS, sig.W, 128)
    s = orc.stft(x, ham, sig.HOP)
    close(s[: sig.W // 2 + 1], g[f"{name}_stft"])
    close(orc.istft(s, ham, sig.HOP), g[f"{name}_istft"])
    close(orc.melspectrogram(x, ham, sig.HOP, fb), g[f"{name}_mel"])
    # the log of a band that holds nothing but the transform's round-off (DC, a tone exactly on a bin) is not reproducible to
    # 1e-12 even between two float64 programs: hold the coefficients to what the reference's own rounding allows
    ref = g[f"{name}_mfcc"]
    floor = mfcc_floor(g[f"{name}_stft"], fb.toarray(), 20, 8.0, np.finfo(np.float64).eps)
    assert excess(orc.mfcc(x, ham, sig.HOP, fb, 20), ref, TOL * np.abs(ref).max() + floor) <= 1.0
    m = orc.mdct(x, kbd)
    close(m, g[f"{name}_mdct"])
    close(orc.imdct(m, kbd), g[f"{name}_imdct"])


@pytest.mark.timeout(300)
def test_signals_that_are_not_noise_cqt(golden):
    import signals as sig
    g = golden["signals"]
    ck = orc.cqtkernel(sig.FS, 24, 55, 3520)
    for name in sig.NAMES:
        xq = sig.signal(name, sig.N_CQT).astype(np.float64)
        assert xq.sum() == g[f"{name}_xq_sum"] and np.abs(xq).sum() == g[f"{name}_xq_abs"]
        close(orc.cqtspectrogram(xq, sig.FS, 25, ck), g[f"{name}_cqt"])
        close(orc.cqtchromagram(xq, sig.FS, 25, 24, ck), g[f"{name}_chroma"])
