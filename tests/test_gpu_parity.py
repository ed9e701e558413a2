"""GPU parity tests (-m gpu): HIP kernels, called through the C-ABI, against the CPU
oracle and the golden vectors produced by the real reference.

Tolerances (BASELINE.json north_star): normwise max|out-ref|/max|ref| per
clip <= 1e-5 for stft/istft/mdct/imdct and <= 1e-4 for mel/mfcc/cqt; MDCT round trip
max|x - imdct(mdct(x))| < 1e-5 absolute on sigma=1 noise; shapes / frame indexing exact.
The reference value is float64 computed from the float32-rounded input.
"""
import numpy as np
import pytest
import scipy.sparse

from conftest import relerr, synth_clip
from oracle import zaf_oracle as orc

pytestmark = pytest.mark.gpu

TOL_FFT = 1e-5
TOL_FB = 1e-4


@pytest.fixture(scope="module")
def zafx():
    """The tests of this module name the kernel a geometry must reach (last_kernel): they run the *_batch functions on COMPACT device arrays,
    the reference's own memory order (set_row_padding("compact")) -- the default since round 6 pads rows off the 128-byte grid, which is
    test_batch_functions_pad_rows_off_the_line_grid's subject."""
    import zafx as z
    assert z.device_count() >= 1
    z.set_row_padding("compact")
    yield z
    z.set_row_padding("auto")


def csr(g, tag):
    return scipy.sparse.csr_matrix((g[f"{tag}_data"], g[f"{tag}_indices"], g[f"{tag}_indptr"]), shape=tuple(g[f"{tag}_shape"]))


# ------------------------------------------------------------------ tiny goldens (W=64)
@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000])
@pytest.mark.parametrize("hop", [32, 16])
def test_tiny_stft_family(zafx, golden, n, hop):
    g = golden["tiny"]
    x, ham = g[f"x_{n}"], g["ham"]
    fb = scipy.sparse.csr_matrix(g["fb_dense"])
    ref = g[f"stft_{n}_{hop}"]
    got = zafx.stft(x, ham, hop)
    assert got.dtype == np.complex128 and got.shape == ref.shape
    assert relerr(got, ref) <= TOL_FFT
    got_tf = zafx.stft_batch(x[None].astype(np.float32), ham, hop, layout="TF")[0]
    assert relerr(got_tf.T, ref) <= TOL_FFT
    y = zafx.istft(ref, ham, hop)
    assert y.dtype == np.float64
    assert relerr(y, g[f"istft_{n}_{hop}"]) <= TOL_FFT
    y_tf = zafx.istft_batch(np.ascontiguousarray(ref.T)[None], ham, hop, layout="TF")[0]
    assert relerr(y_tf, g[f"istft_{n}_{hop}"]) <= TOL_FFT
    assert relerr(zafx.melspectrogram(x, ham, hop, fb), g[f"mel_{n}_{hop}"]) <= TOL_FB
    assert relerr(zafx.mfcc(x, ham, hop, fb, 5), g[f"mfcc_{n}_{hop}"]) <= TOL_FB


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000])
@pytest.mark.parametrize("wname", ["sine", "kbd"])
def test_tiny_mdct_family(zafx, golden, n, wname):
    g = golden["tiny"]
    x, w = g[f"x_{n}"], g[wname]
    ref = g[f"mdct_{wname}_{n}"]
    got = zafx.mdct(x, w)
    assert got.shape == ref.shape and relerr(got, ref) <= TOL_FFT
    got_tf = zafx.mdct_batch(x[None].astype(np.float32), w, layout="TF")[0]
    assert relerr(got_tf.T, ref) <= TOL_FFT
    yref = g[f"imdct_{wname}_{n}"]
    y = zafx.imdct(ref, w)
    assert y.shape == yref.shape and relerr(y, yref) <= TOL_FFT
    y_tf = zafx.imdct_batch(np.ascontiguousarray(ref.T)[None], w, layout="TF")[0]
    assert relerr(y_tf, yref) <= TOL_FFT


def test_tiny_istft_generic(zafx, golden):
    """istft of a NON-Hermitian spectrum: the reference takes real(ifft(.)) (zaf.py:223)."""
    g = golden["tiny"]
    y = zafx.istft(g["istft_generic_in"], g["ham"], 32)
    assert relerr(y, g["istft_generic_out"]) <= TOL_FFT


@pytest.mark.parametrize("n", [400, 4000, 4321])
def test_tiny_cqt(zafx, golden, n):
    g = golden["tiny"]
    ck = scipy.sparse.csr_matrix(g["ck_dense"])
    x = g[f"xq_{n}"]
    got = zafx.cqtspectrogram(x, 4000, 50, ck)
    assert got.shape == g[f"cqt_{n}"].shape and relerr(got, g[f"cqt_{n}"]) <= TOL_FB
    got = zafx.cqtchromagram(x, 4000, 50, 12, ck)
    assert got.shape == g[f"chroma_{n}"].shape and relerr(got, g[f"chroma_{n}"]) <= TOL_FB
    got_tf = zafx.cqtspectrogram_batch(x[None].astype(np.float32), 4000, 50, ck, layout="TF")[0]
    assert relerr(got_tf.T, g[f"cqt_{n}"]) <= TOL_FB


# ------------------------------------------------------------------ other sizes vs the oracle
@pytest.mark.parametrize("wl,hop,n", [(128, 64, 777), (256, 64, 3000), (512, 256, 5000), (1024, 512, 9000),
                                       (2048, 512, 20000), (2048, 1024, 1), (2048, 1024, 2049), (4096, 1024, 30000),
                                       (8192, 4096, 50000), (2048, 700, 12345)])
def test_stft_sizes(zafx, wl, hop, n):
    x = np.stack([synth_clip(5, c, n) for c in range(3)])
    w = zafx.hamming(wl)
    ref = orc.stft_batch(x.astype(np.float64), w, hop)
    for layout in ("FT", "TF"):
        got = zafx.stft_batch(x, w, hop, layout=layout)
        if layout == "TF":
            got = got.transpose(0, 2, 1)
        assert got.shape == ref.shape
        for c in range(3):
            assert relerr(got[c], ref[c]) <= TOL_FFT, (layout, c)
    for c in range(3):
        yref = orc.istft(ref[c], w, hop)
        y = zafx.istft_batch(ref[c][None], w, hop)[0]
        assert y.shape == yref.shape and relerr(y, yref) <= TOL_FFT


@pytest.mark.parametrize("wl,n", [(128, 777), (256, 3000), (512, 5000), (1024, 9000), (2048, 1), (2048, 2049),
                                   (2048, 20000), (4096, 30000), (8192, 50000)])
def test_mdct_sizes(zafx, wl, n):
    x = np.stack([synth_clip(6, c, n) for c in range(3)])
    w = zafx.kaiser_bessel_derived(wl)
    ref = orc.mdct_batch(x.astype(np.float64), w)
    for layout in ("FT", "TF"):
        got = zafx.mdct_batch(x, w, layout=layout)
        if layout == "TF":
            got = got.transpose(0, 2, 1)
        assert got.shape == ref.shape
        for c in range(3):
            assert relerr(got[c], ref[c]) <= TOL_FFT, (layout, c)
    yref = orc.imdct(ref[0], w)
    y = zafx.imdct_batch(ref[:1], w)[0]
    assert y.shape == yref.shape and relerr(y, yref) <= TOL_FFT
    k = min(n, len(y))
    assert np.max(np.abs(y[:k] - x[0, :k])) < 1e-5


@pytest.mark.parametrize("n,clips", [(30000, 3), (4, 1), (2048 * 31, 2), (2048 * 32 + 4, 2), (2048 * 70, 40), (441000, 2), (30002, 2)])
def test_mdct_w4096_two_bands(zafx, n, clips):
    """W = 4096 in the reference layout runs k_mdct_ft32b (32-frame tiles as two bands of bins; zaf.py:1047-1073 is what it
    replaces) for clips of a multiple of four samples, the generic kernel otherwise: ragged and whole last tiles, rows on and off
    the line grid, more tiles than workgroups, padded rows."""
    x = np.stack([synth_clip(43, c % 5, n) for c in range(clips)])
    w = zafx.kaiser_bessel_derived(4096)
    assert zafx.mdct_plan(w).kernel_name == "k_mdct_ft32b"   # planned: the band family; what RAN depends on the call (last_kernel)
    ref = orc.mdct_batch(x[:5].astype(np.float64), w)
    got = zafx.mdct_batch(x, w)
    T = got.shape[2]
    # clips of a multiple of four samples ride the bands -- rows off the 64-byte grid on the carry form --, others the generic kernel
    assert zafx.mdct_plan(w).last_kernel == ("k_mdct" if n % 4 else "k_mdct_ft32b" if T % 16 == 0 else "k_mdct_ft32bc"), (n, T)
    assert got.shape[1:] == ref.shape[1:] and got.dtype == np.float32
    for c in range(clips):
        assert relerr(got[c], ref[c % 5]) <= TOL_FFT, c
    y = zafx.imdct_batch(got[:1], w)[0]
    k = min(n, len(y))
    assert np.max(np.abs(y[:k] - x[0, :k])) < 1e-5
    _run_padded(zafx, zafx.mdct_plan(w), zafx.mdct_plan(w, row_align=32), None, None, x[:2])


@pytest.mark.parametrize("n,clips", [(441000, 3), (4096 * 31, 2), (4096 * 30 + 1, 2), (70001, 3), (1, 1), (4096 * 63, 40), (4096 * 95 - 7, 90), (8193, 2)])
def test_mdct_w8192_four_bands(zafx, n, clips):
    """W = 8192 in the reference layout runs k_mdct_ft32q (32-frame tiles as four bands of bins, the frame folded twice; zaf.py:1047-1073 is
    what it replaces): any clip length (4-byte loads), ragged and whole last tiles, rows on and off the line grid, even and odd T (frame
    pairs as 8-byte stores or single values), more tiles than workgroups, padded rows; the other layout stays on the generic kernel."""
    x = np.stack([synth_clip(45, c % 5, n) for c in range(clips)])
    w = zafx.kaiser_bessel_derived(8192)
    assert zafx.mdct_plan(w).kernel_name == "k_mdct_ft32q"
    ref = orc.mdct_batch(x[:5].astype(np.float64), w)
    got = zafx.mdct_batch(x, w)
    assert zafx.mdct_plan(w).last_kernel == "k_mdct_ft32q"
    assert got.shape[1:] == ref.shape[1:] and got.dtype == np.float32
    for c in range(clips):
        assert relerr(got[c], ref[c % 5]) <= TOL_FFT, c
    y = zafx.imdct_batch(got[:1], w)[0]
    k = min(n, len(y))
    assert np.max(np.abs(y[:k] - x[0, :k])) < 1e-5
    _run_padded(zafx, zafx.mdct_plan(w), zafx.mdct_plan(w, row_align=32), None, None, x[:2])
    tf = zafx.mdct_batch(x[:2], w, layout="TF")
    assert zafx.mdct_plan(w, layout="TF").last_kernel == "k_mdct"
    for c in range(len(tf)):
        assert relerr(tf[c].T, ref[c]) <= TOL_FFT


def test_imdct_batches_every_clip_and_tile_shape(zafx):
    """The inverse of a BATCH, every clip compared: the output length (T-1) M - 1 is odd, so every second clip starts on a 4-byte
    boundary (the sweep form of the overlap-add stores those as two 4-byte values); lengths that end on a full tile, a partial
    tile, a single tile and a single frame pair; both layouts."""
    w = zafx.kaiser_bessel_derived(2048)
    for n in (1, 1024 * 31, 1024 * 32, 1024 * 33 + 5, 1024 * 64, 70001):
        x = np.stack([synth_clip(61, c, n) for c in range(5)])
        coefs = orc.mdct_batch(x.astype(np.float64), w)
        for layout in ("FT", "TF"):
            c_in = coefs if layout == "FT" else np.ascontiguousarray(coefs.transpose(0, 2, 1))
            y = zafx.imdct_batch(c_in, w, layout=layout)
            for c in range(5):
                yref = orc.imdct(coefs[c], w)
                assert y[c].shape == yref.shape and relerr(y[c], yref) <= TOL_FFT, (n, layout, c)


# ------------------------------------------------------------------ BASELINE config clips (full clip size)
@pytest.fixture(scope="module")
def config_S():
    x = np.stack([synth_clip(0, c, 441000) for c in range(2)])
    return x


def _check_probes(g, key, arr, tol):
    assert tuple(g[f"{key}_shape"]) == arr.shape
    flat = arr.reshape(-1)
    scale = float(g[f"{key}_maxabs"])
    assert np.max(np.abs(flat[g[f"{key}_idx"]] - g[f"{key}_val"])) <= tol * scale


def test_config_stft_istft(zafx, golden, config_S):
    g = golden["config"]
    ham = zafx.hamming(2048)
    got = zafx.stft_batch(config_S, ham, 1024)
    assert got.shape == (2, 2048, 432) and got.dtype == np.complex64
    for c in range(2):
        _check_probes(g, f"S{c}_stft", got[c].astype(np.complex128), TOL_FFT)
        ref = orc.stft(config_S[c].astype(np.float64), ham, 1024)
        assert relerr(got[c], ref) <= TOL_FFT
        # Hermitian mirror, exactly-real DC / Nyquist
        assert np.array_equal(got[c][1:1024], np.conj(got[c][:1024:-1]))
        assert np.all(got[c][0].imag == 0) and np.all(got[c][1024].imag == 0)
        y = zafx.istft_batch(ref[None], ham, 1024)[0]
        _check_probes(g, f"S{c}_istft", y.astype(np.float64), TOL_FFT)
        assert np.max(np.abs(y[:441000] - config_S[c])) < 1e-5   # COLA resynthesis (zaf.py:165-194)


def test_config_mel_mfcc(zafx, golden, config_S):
    g = golden["config"]
    ham = zafx.hamming(2048)
    fb = zafx.melfilterbank(44100, 2048, 128)
    assert relerr(fb.toarray(), csr(golden["consts"], "fb128").toarray()) == 0.0
    mel = zafx.melspectrogram_batch(config_S, ham, 1024, fb)
    mf = zafx.mfcc_batch(config_S, ham, 1024, fb, 20)
    assert mel.shape == (2, 128, 432) and mf.shape == (2, 20, 432)
    for c in range(2):
        _check_probes(g, f"S{c}_mel", mel[c].astype(np.float64), TOL_FB)
        _check_probes(g, f"S{c}_mfcc", mf[c].astype(np.float64), TOL_FB)
        x64 = config_S[c].astype(np.float64)
        assert relerr(mel[c], orc.melspectrogram(x64, ham, 1024, fb)) <= TOL_FB
        assert relerr(mf[c], orc.mfcc(x64, ham, 1024, fb, 20)) <= TOL_FB
    mel_tf = zafx.melspectrogram_batch(config_S, ham, 1024, fb, layout="TF")
    assert np.array_equal(mel_tf.transpose(0, 2, 1), mel)
    fb40 = zafx.melfilterbank(44100, 2048, 40)
    mf40 = zafx.mfcc_batch(config_S[:1], ham, 1024, fb40, 13)[0]
    assert relerr(mf40, orc.mfcc(config_S[0].astype(np.float64), ham, 1024, fb40, 13)) <= TOL_FB


def test_config_mdct_roundtrip(zafx, golden, config_S):
    g = golden["config"]
    kbd = zafx.kaiser_bessel_derived(2048)
    m = zafx.mdct_batch(config_S, kbd)
    assert m.shape == (2, 1024, 432)
    for c in range(2):
        _check_probes(g, f"S{c}_mdct", m[c].astype(np.float64), TOL_FFT)
        assert relerr(m[c], orc.mdct(config_S[c].astype(np.float64), kbd)) <= TOL_FFT
    y = zafx.imdct_batch(m, kbd)
    assert y.shape == (2, 441343)
    for c in range(2):
        _check_probes(g, f"S{c}_imdct", y[c].astype(np.float64), TOL_FFT)
        assert np.max(np.abs(y[c][:440999] - config_S[c][:440999])) < 1e-5   # BASELINE config 4 residual


def test_config_cqt(zafx, golden):
    g = golden["config"]
    ck = csr(golden["consts"], "ck")
    x = synth_clip(0, 0, 1323000)
    got = zafx.cqtspectrogram_batch(x[None], 44100, 25, ck)[0]
    assert got.shape == (144, 750)
    _check_probes(g, "Q0_cqt", got.astype(np.float64), TOL_FB)
    ch = zafx.cqtchromagram_batch(x[None], 44100, 25, 24, ck)[0]
    _check_probes(g, "Q0_chroma", ch.astype(np.float64), TOL_FB)
    # shorter clip, full comparison with the oracle (and our own kernel builder)
    ck2 = zafx.cqtkernel(44100, 24, 55, 3520)
    xs = x[:200000]
    ref = orc.cqtspectrogram(xs.astype(np.float64), 44100, 25, ck2)
    got = zafx.cqtspectrogram(xs, 44100, 25, ck2)
    assert got.shape == ref.shape and relerr(got, ref) <= TOL_FB


def test_cqt_reference_docstring_kernel(zafx, golden):
    """cqtkernel's own docstring example (zaf.py:476-483): f_max = fs/2 -> 208 bins, 60 879 non-zeros, columns up to 16 613
    (upper half of the spectrum: conjugates of the one-sided bins).  Reference golden, drop-in signatures."""
    g = golden["cqtfull"]
    ck = zafx.cqtkernel(44100, 24, 55, 44100 / 2)
    assert ck.shape == (208, 32768) and ck.nnz == 60879 and np.array_equal(ck.tocsr().indptr, g["indptr"])
    x = synth_clip(5, 0, 100000)
    got = zafx.cqtspectrogram(x, 44100, 25, ck)
    assert got.shape == g["cqt"].shape == (208, 56) and relerr(got, g["cqt"]) <= TOL_FB
    ch = zafx.cqtchromagram(x, 44100, 25, 24, ck)
    assert ch.shape == (24, 56) and relerr(ch, g["chroma"]) <= TOL_FB
    b = zafx.cqtspectrogram_batch(np.stack([x, synth_clip(5, 1, 100000)]), 44100, 25, ck)
    assert relerr(b[0], g["cqt"]) <= TOL_FB
    assert relerr(b[1], orc.cqtspectrogram(synth_clip(5, 1, 100000).astype(np.float64), 44100, 25, ck)) <= TOL_FB
    # kernels between the benchmark's and this one (168 bins needed more LDS than the float32 kernel had)
    ck2 = zafx.cqtkernel(44100, 24, 55, 7040)
    assert relerr(zafx.cqtspectrogram(x, 44100, 25, ck2), orc.cqtspectrogram(x.astype(np.float64), 44100, 25, ck2)) <= TOL_FB


# ------------------------------------------------------------------ full BASELINE batch, device resident
def test_full_batch_device_resident(zafx):
    """1024 clips x 10 s (BASELINE configs 2 and 4): size-independent properties on the whole
    batch (stft->istft and mdct->imdct round trips, replica consistency), oracle on 4 clips."""
    B, N, W, H = 1024, 441000, 2048, 1024
    distinct = 8
    base = np.stack([synth_clip(0, c, N) for c in range(distinct)])
    d_x = zafx.DeviceBuffer((B, N), np.float32)
    d_base = zafx.DeviceBuffer.from_host(base)
    for r in range(B // distinct):
        d_x.copy_from(d_base, dst_offset=r * distinct * N * 4)
    ham, kbd = zafx.hamming(W), zafx.kaiser_bessel_derived(W)

    fwd, inv = zafx.stft_plan(ham, H), zafx.istft_plan(ham, H)
    T = fwd.out_dims(N)[1]
    d_spec = zafx.DeviceBuffer(fwd.out_shape(B, N), np.complex64)
    d_y = zafx.DeviceBuffer(inv.out_shape(B, T), np.float32)
    fwd.execute(d_x, d_spec, B, N)
    fwd.sync()
    inv.execute(d_spec, d_y, B, T)
    inv.sync()
    first, last = d_spec.download(0, distinct), d_spec.download(B - distinct, distinct)
    assert np.array_equal(first, last)   # replicas of the same clips give identical bits
    for c in range(4):
        assert relerr(first[c], orc.stft(base[c].astype(np.float64), ham, H)) <= TOL_FFT
    y = d_y.download()
    assert y.shape == (B, 441344)
    err = np.max(np.abs(y[:, :N].reshape(B // distinct, distinct, N) - base[None]))
    assert err < 1e-5
    d_spec.free(); d_y.free()

    fwd, inv = zafx.mdct_plan(kbd), zafx.mdct_plan(kbd, inverse=True)
    d_m = zafx.DeviceBuffer(fwd.out_shape(B, N), np.float32)
    d_r = zafx.DeviceBuffer(inv.out_shape(B, T), np.float32)
    fwd.execute(d_x, d_m, B, N)
    fwd.sync()
    inv.execute(d_m, d_r, B, T)
    inv.sync()
    r = d_r.download()
    assert r.shape == (B, 441343)
    err = np.max(np.abs(r[:, :N - 1].reshape(B // distinct, distinct, N - 1) - base[None, :, :N - 1]))
    assert err < 1e-5   # BASELINE config 4: residual < 1e-5
    for buf in (d_m, d_r, d_x, d_base):
        buf.free()


# ------------------------------------------------------------------ error behaviour
def test_errors(zafx):
    ham = zafx.hamming(2048)
    x = synth_clip(1, 0, 5000)
    with pytest.raises(ValueError):
        zafx.stft(np.stack([x, x]), ham, 1024)        # 2-D input (reference breaks at zaf.py:135)
    with pytest.raises(ValueError):
        zafx.stft(x, ham, 1024.0)                     # non-int hop
    with pytest.raises(ValueError):
        zafx.stft(x, zafx.hamming(9000), 500)         # window above 8192
    with pytest.raises(ValueError):
        zafx.melspectrogram(x, ham, 1024, np.ones((4, 1024)))   # filterbank must expose .toarray()
    with pytest.raises(zafx.ZafxError):                          # a window_length the C-ABI has no kernel for
        zafx.Plan(zafx.ISTFT, window_length=9000, step_length=64)
    with pytest.raises(zafx.ZafxError):                          # float64 Bluestein forms stop at 2048 samples
        zafx.Plan(zafx.STFT, window_length=3000, step_length=500, f64=True)


# ------------------------------------------------------------------ RCCL broadcast of constants (1 rank)
def test_rccl_broadcast_single_rank(zafx, golden):
    """zafx_comm_* end to end with n_ranks = 1: communicator creation, header + payload
    broadcast of every constant kind, re-packing, and a transform afterwards."""
    comm = zafx.Comm(0, 0, 1, zafx.Comm.unique_id())
    ham = zafx.hamming(2048)
    fb = zafx.melfilterbank(44100, 2048, 128)
    x = synth_clip(2, 0, 30000)
    plan = zafx.mel_plan(ham, 1024, fb, 20)
    before = plan.run_host(x[None], len(x))
    comm.broadcast_constants(plan, root=0)
    after = plan.run_host(x[None], len(x))
    assert np.array_equal(before, after)
    ck = scipy.sparse.csr_matrix(golden["tiny"]["ck_dense"])
    qplan = zafx.cqt_plan(4000, 50, ck)
    xq = golden["tiny"]["xq_4000"].astype(np.float32)
    before = qplan.run_host(xq[None], len(xq))
    comm.broadcast_constants(qplan, root=0)
    assert np.array_equal(before, qplan.run_host(xq[None], len(xq)))
    comm.destroy()


# ------------------------------------------------------------------ unaligned / odd geometries (slow-path kernels)
@pytest.mark.parametrize("n,hop", [(44101, 1024), (44100, 1023), (30001, 511), (5000, 2048)])
def test_stft_family_unaligned(zafx, n, hop):
    """Odd clip lengths / odd hops take the predicated (non 8-byte-aligned) load path."""
    x = np.stack([synth_clip(8, c, n) for c in range(5)])
    ham = zafx.hamming(2048)
    ref = orc.stft_batch(x.astype(np.float64), ham, hop)
    got = zafx.stft_batch(x, ham, hop)
    assert got.shape == ref.shape
    for c in range(5):
        assert relerr(got[c], ref[c]) <= TOL_FFT
    fb = zafx.melfilterbank(44100, 2048, 128)
    mel = zafx.melspectrogram_batch(x[:2], ham, hop, fb)
    mf = zafx.mfcc_batch(x[:2], ham, hop, fb, 20)
    for c in range(2):
        x64 = x[c].astype(np.float64)
        assert relerr(mel[c], orc.melspectrogram(x64, ham, hop, fb)) <= TOL_FB
        assert relerr(mf[c], orc.mfcc(x64, ham, hop, fb, 20)) <= TOL_FB
    if -(-2048 // hop) <= 8:
        y = zafx.istft_batch(ref[:2], ham, hop)
        for c in range(2):
            assert relerr(y[c], orc.istft(ref[c], ham, hop)) <= TOL_FFT


@pytest.mark.parametrize("n", [1024 * 30, 1024 * 31 + 1, 1024 * 33 - 1, 100000])
def test_mdct_frame_count_parity(zafx, n):
    """Even and odd frame counts (the 8-byte pair store needs T even), partial last tile of 32."""
    x = np.stack([synth_clip(9, c, n) for c in range(3)])
    w = zafx.kaiser_bessel_derived(2048)
    ref = orc.mdct_batch(x.astype(np.float64), w)
    got = zafx.mdct_batch(x, w)
    assert got.shape == ref.shape
    for c in range(3):
        assert relerr(got[c], ref[c]) <= TOL_FFT
    y = zafx.imdct_batch(got, w)
    for c in range(3):
        k = min(n, y.shape[1])
        assert np.max(np.abs(y[c][:k] - x[c][:k])) < 1e-5


@pytest.mark.parametrize("wl,hop,fs,nmel,ncoef", [(1024, 256, 22050, 64, 13), (512, 128, 16000, 40, 13), (256, 100, 8000, 26, 12),
                                                  (128, 64, 8000, 20, 10), (512, 512, 16000, 80, 20)])
def test_mel_other_window(zafx, wl, hop, fs, nmel, ncoef):
    """The other instantiations of the fused kernel (W = 128 ... 1024: slots in the upper halves of smaller frame buffers
    or in their own region, several frames per wavefront below W = 128 x 64 lanes)."""
    x = np.stack([synth_clip(10, c, 22050 + 7 * c) [:22050] for c in range(2)])
    w = zafx.hamming(wl)
    fb = zafx.melfilterbank(fs, wl, nmel)
    for layout in ("FT", "TF"):
        mel = zafx.melspectrogram_batch(x, w, hop, fb, layout=layout)
        mf = zafx.mfcc_batch(x, w, hop, fb, ncoef, layout=layout)
        if layout == "TF":
            mel, mf = mel.transpose(0, 2, 1), mf.transpose(0, 2, 1)
        for c in range(2):
            x64 = x[c].astype(np.float64)
            ref_mel, ref_mf = orc.melspectrogram(x64, w, hop, fb), orc.mfcc(x64, w, hop, fb, ncoef)
            assert mel[c].shape == ref_mel.shape and mf[c].shape == ref_mf.shape
            assert relerr(mel[c], ref_mel) <= TOL_FB, (layout, c)
            assert relerr(mf[c], ref_mf) <= TOL_FB, (layout, c)


def _replicated_on_device(zafx, base, B):
    """(B, N) float32 device array holding base[c % len(base)] at clip c."""
    distinct, N = base.shape
    d_x = zafx.DeviceBuffer((B, N), np.float32)
    d_base = zafx.DeviceBuffer.from_host(base)
    for r in range(B // distinct):
        d_x.copy_from(d_base, dst_offset=r * distinct * N * 4)
    d_base.free()
    return d_x


def test_full_batch_mel_mfcc_device_resident(zafx):
    """BASELINE config 3 at the size bench.py runs it: 1024 clips x 10 s = 27 648 tiles, so every persistent k_mel
    workgroup walks > 100 tiles (the tile-to-tile LDS reuse).  8 distinct clips; first, middle and last replicas
    against the oracle and bit-equal among themselves."""
    B, N, W, H, distinct = 1024, 441000, 2048, 1024, 8
    base = np.stack([synth_clip(0, c, N) for c in range(distinct)])
    d_x = _replicated_on_device(zafx, base, B)
    ham, fb = zafx.hamming(W), zafx.melfilterbank(44100, W, 128)
    for ncoef, rows in ((None, 128), (20, 20)):
        plan = zafx.mel_plan(ham, H, fb, ncoef)
        d_out = zafx.DeviceBuffer(plan.out_shape(B, N), np.float32)
        plan.execute(d_x, d_out, B, N)
        plan.sync()
        first, mid, last = (d_out.download(s, distinct) for s in (0, B // 2, B - distinct))
        assert first.shape == (distinct, rows, 432)
        assert np.array_equal(first, mid) and np.array_equal(first, last)
        for c in range(distinct):
            x64 = base[c].astype(np.float64)
            ref = orc.melspectrogram(x64, ham, H, fb) if ncoef is None else orc.mfcc(x64, ham, H, fb, ncoef)
            assert relerr(first[c], ref) <= TOL_FB
        # the whole batch: every replica group identical (no tile of any workgroup's walk differs)
        whole = d_out.download().reshape(B // distinct, distinct, rows, 432)
        assert np.array_equal(whole, np.broadcast_to(first[None], whole.shape))
        d_out.free()
    d_x.free()


def test_full_batch_mel_mfcc_one_pass_device_resident(zafx):
    """BASELINE config 3 as ONE pass (zaf.melspectrogram and zaf.mfcc start with the same zaf.stft, zaf.py:369 / :436): the with_mel form of the
    mfcc plan (k_mel2 MODE 4) on the 1024-clip batch -- rows 0..127 the melspectrogram, rows 128..147 the MFCCs, both BIT-EQUAL to the
    single-output plans' results over the whole batch."""
    B, N, W, H, distinct = 1024, 441000, 2048, 1024, 8
    base = np.stack([synth_clip(0, c, N) for c in range(distinct)])
    d_x = _replicated_on_device(zafx, base, B)
    ham, fb = zafx.hamming(W), zafx.melfilterbank(44100, W, 128)
    singles = []
    for ncoef in (None, 20):
        plan = zafx.mel_plan(ham, H, fb, ncoef)
        d_out = zafx.DeviceBuffer(plan.out_shape(B, N), np.float32)
        plan.execute(d_x, d_out, B, N)
        plan.sync()
        singles.append(d_out.download())
        d_out.free()
    both = zafx.mel_plan(ham, H, fb, 20, also_mel=True)
    assert both.out_shape(B, N) == (B, 148, 432)
    d_out = zafx.DeviceBuffer(both.out_shape(B, N), np.float32)
    both.execute(d_x, d_out, B, N)
    both.sync()
    assert both.last_kernel == "k_mel2"
    got = d_out.download()
    assert np.array_equal(got[:, :128], singles[0]) and np.array_equal(got[:, 128:], singles[1])
    d_out.free()
    d_x.free()


@pytest.mark.parametrize("hop,n,clips,nmel,ncoef", [(1024, 100000, 3, 128, 20), (512, 30001, 2, 128, 13), (777, 20000, 2, 40, 32), (1024, 1, 2, 64, 20),
                                                     (2048, 50000, 5, 13, 12)])
def test_mel_mfcc_one_pass(zafx, hop, n, clips, nmel, ncoef):
    """mel_mfcc_batch: both outputs bit-equal to melspectrogram_batch / mfcc_batch (and so within 1e-4 of the reference), whole tiles, edge
    tiles, odd hops and lengths (the unaligned form), filterbanks whose blocks are cut in several parts, both layouts, padded rows, int16 PCM
    in the kernel's loads (mono and stereo); geometries outside the one-pass kernel run the two plans."""
    x = np.stack([synth_clip(59, c % 7, n) for c in range(clips)])
    w = zafx.hamming(2048)
    fb = zafx.melfilterbank(44100, 2048, nmel)
    assert zafx.mel_mfcc_supported(2048, nmel, ncoef)
    for layout in ("FT", "TF"):
        mel, cep = zafx.mel_mfcc_batch(x, w, hop, fb, ncoef, layout=layout)
        assert zafx.mel_plan(w, hop, fb, ncoef, layout=layout, also_mel=True).last_kernel == "k_mel2"
        assert np.array_equal(mel, zafx.melspectrogram_batch(x, w, hop, fb, layout=layout))
        assert np.array_equal(cep, zafx.mfcc_batch(x, w, hop, fb, ncoef, layout=layout))
    for c in range(clips):
        x64 = x[c].astype(np.float64)
        assert relerr(mel[c].T, orc.melspectrogram(x64, w, hop, fb)) <= TOL_FB and relerr(cep[c].T, orc.mfcc(x64, w, hop, fb, ncoef)) <= TOL_FB
    # padded rows through the plan interface
    pl = zafx.mel_plan(w, hop, fb, ncoef, row_align=32, also_mel=True)
    got = pl.run_host(x, n)
    mel_ft, cep_ft = zafx.mel_mfcc_batch(x, w, hop, fb, ncoef)
    assert np.array_equal(got[:, :nmel], mel_ft) and np.array_equal(got[:, nmel:], cep_ft)
    # int16 PCM, one and two channels
    pcm = np.clip(np.rint(x * 3000.0), -32768, 32767).astype(np.int16)
    for p in (pcm[:, :, None], np.stack([pcm, pcm[::-1]], axis=-1)):
        p = np.ascontiguousarray(p)
        mel_p, cep_p = zafx.mel_mfcc_pcm_batch(p, w, hop, fb, ncoef)
        assert np.array_equal(mel_p, zafx.melspectrogram_pcm_batch(p, w, hop, fb)) and np.array_equal(cep_p, zafx.mfcc_pcm_batch(p, w, hop, fb, ncoef))


def test_mel_mfcc_outside_the_one_pass_kernel(zafx):
    """Other windows, more coefficients, float64: mel_mfcc_batch runs the two plans (same results); the plan factory and the C-ABI refuse."""
    x = np.stack([synth_clip(61, c, 20000) for c in range(2)])
    for wl, nmel, ncoef, f64 in ((1024, 64, 13, False), (2048, 128, 40, False), (4096, 128, 20, False), (2048, 128, 20, True)):
        w, fb = zafx.hamming(wl), zafx.melfilterbank(44100, wl, nmel)
        assert not zafx.mel_mfcc_supported(wl, nmel, ncoef, f64)
        mel, cep = zafx.mel_mfcc_batch(x, w, wl // 2, fb, ncoef, f64=f64)
        assert np.array_equal(mel, zafx.melspectrogram_batch(x, w, wl // 2, fb, f64=f64)) and np.array_equal(cep, zafx.mfcc_batch(x, w, wl // 2, fb, ncoef, f64=f64))
        with pytest.raises(ValueError):
            zafx.mel_plan(w, wl // 2, fb, ncoef, also_mel=True, f64=f64)
    with pytest.raises(zafx.ZafxError):
        zafx.Plan(zafx.MFCC, window_length=1024, step_length=512, n_filters=64, n_coefs=13, with_mel=True)
    with pytest.raises(zafx.ZafxError):
        zafx.Plan(zafx.MEL, window_length=2048, step_length=1024, n_filters=64, with_mel=True)


@pytest.mark.parametrize("n,clips", [(4096 * 11 + 100, 2), (4096 * 30, 3), (5000, 1), (4096 * 70 + 1, 5), (4096 * 15, 300)])
def test_istft_w8192_four_classes(zafx, n, clips):
    """W = 8192, hop 4096 in the reference layout: k_istft_ft8q (round 6: the rows class by class -- 4q, 8p + 2, 8p + 1, 8p + 5 and their mirrors --,
    one 1024-point inverse per class and frame, the quarter frames out of butterflies in registers; zaf.py:223 and :226-241 are what it
    replaces) -- odd and even frame counts (8-byte and 16-byte row pieces), one and many tiles per clip, clips cut into segments, one-sided
    input, an input that is not Hermitian, padded rows; other hops stay on the generic kernel."""
    x = np.stack([synth_clip(73, c % 5, n) for c in range(clips)])
    w = zafx.hamming(8192)
    spec = np.stack([orc.stft(x[c].astype(np.float64), w, 4096) for c in range(min(clips, 5))])
    spec = spec[np.arange(clips) % spec.shape[0]].astype(np.complex64)
    ref = [orc.istft(spec[c].astype(np.complex128), w, 4096) for c in range(min(clips, 5))]
    for one in (False, True):
        y = zafx.istft_batch(np.ascontiguousarray(spec[:, :4097] if one else spec), w, 4096, onesided=one)
        assert zafx.istft_plan(w, 4096, onesided=one).last_kernel == "k_istft_ft8q"
        assert y.shape == (clips, len(ref[0]))
        for c in range(clips):
            assert relerr(y[c], ref[c % 5]) <= TOL_FFT, (one, c)
        k = min(n, y.shape[1])
        assert np.max(np.abs(y[:, :k] - x[:, :k])) < 1e-5   # COLA resynthesis (zaf.py:165-194)
    rng = np.random.default_rng(n)
    noisy = (spec[:2] + 0.05 * (rng.standard_normal(spec[:2].shape) + 1j * rng.standard_normal(spec[:2].shape))).astype(np.complex64)
    y = zafx.istft_batch(noisy, w, 4096)
    for c in range(len(noisy)):
        assert relerr(y[c], orc.istft(noisy[c].astype(np.complex128), w, 4096)) <= TOL_FFT   # real(ifft(.)) of anything (zaf.py:223)
    pl = zafx.istft_plan(w, 4096, row_align=16)
    assert relerr(pl.run_host(spec[:1], spec.shape[2]), np.stack(ref[:1])) <= TOL_FFT and pl.last_kernel == "k_istft_ft8q"
    zafx.istft_batch(np.stack([orc.stft(x[0].astype(np.float64), w, 2048)]).astype(np.complex64), w, 2048)
    assert zafx.istft_plan(w, 2048).last_kernel != "k_istft_ft8q"


@pytest.mark.parametrize("n,clips", [(2048 * 11 + 100, 2), (2048 * 31, 3), (3000, 1), (2048 * 70 + 1, 5), (2048 * 40, 300), (441000, 9)])
def test_istft_w4096_two_classes(zafx, n, clips):
    """W = 4096, hop 2048 in the reference layout: k_istft_ft16d (round 6: the even rows as the W = 2048 inverse of x0 + x1, the rows 4p + 1 and their
    mirrors as one 1024-point inverse that gives x0 - x1; every row read once where k_istft_ft16b read every row per band; zaf.py:223 and :226-241
    are what it replaces) -- odd and even frame counts (8-byte and 16-byte row pieces), one and many tiles per clip, clips cut into segments,
    one-sided input, an input that is not Hermitian, padded rows; other hops stay on the band kernel."""
    x = np.stack([synth_clip(83, c % 5, n) for c in range(clips)])
    w = zafx.hamming(4096)
    spec = np.stack([orc.stft(x[c].astype(np.float64), w, 2048) for c in range(min(clips, 5))])
    spec = spec[np.arange(clips) % spec.shape[0]].astype(np.complex64)
    ref = [orc.istft(spec[c].astype(np.complex128), w, 2048) for c in range(min(clips, 5))]
    for one in (False, True):
        y = zafx.istft_batch(np.ascontiguousarray(spec[:, :2049] if one else spec), w, 2048, onesided=one)
        assert zafx.istft_plan(w, 2048, onesided=one).last_kernel == "k_istft_ft16d"
        assert y.shape == (clips, len(ref[0]))
        for c in range(clips):
            assert relerr(y[c], ref[c % 5]) <= TOL_FFT, (one, c)
        k = min(n, y.shape[1])
        assert np.max(np.abs(y[:, :k] - x[:, :k])) < 1e-5   # COLA resynthesis (zaf.py:165-194)
    rng = np.random.default_rng(n)
    noisy = (spec[:2] + 0.05 * (rng.standard_normal(spec[:2].shape) + 1j * rng.standard_normal(spec[:2].shape))).astype(np.complex64)
    y = zafx.istft_batch(noisy, w, 2048)
    for c in range(len(noisy)):
        assert relerr(y[c], orc.istft(noisy[c].astype(np.complex128), w, 2048)) <= TOL_FFT   # real(ifft(.)) of anything (zaf.py:223)
    pl = zafx.istft_plan(w, 2048, row_align=16)
    assert relerr(pl.run_host(spec[:1], spec.shape[2]), np.stack(ref[:1])) <= TOL_FFT and pl.last_kernel == "k_istft_ft16d"
    zafx.istft_batch(np.stack([orc.stft(x[0].astype(np.float64), w, 1024)]).astype(np.complex64), w, 1024)
    assert zafx.istft_plan(w, 1024).last_kernel == "k_istft_ft16b"


@pytest.mark.parametrize("n,clips", [(4096 * 11 + 100, 2), (4096 * 30, 3), (5000, 1), (4096 * 70 + 1, 5), (4096 * 15, 300)])
def test_imdct_w8192_two_classes(zafx, n, clips):
    """W = 8192 in the reference layout: k_imdct_q (round 6: the coefficient rows as two classes by the parity of m -- 4m', M-1-4m' and 4m'+2,
    M-3-4m' --, one 1024-point transform per class and frame, joined in registers; zaf.py:1138-1182 is what it replaces) -- frame counts on and
    off the 16-byte grid of the rows (off it the batch function pads; the compact array stays on the generic kernel), one and many tiles per
    clip, clips cut into segments, coefficients that are no MDCT of anything, TDAC reconstruction."""
    x = np.stack([synth_clip(79, c % 5, n) for c in range(clips)])
    w = zafx.kaiser_bessel_derived(8192)
    co = np.stack([orc.mdct(x[c].astype(np.float64), w) for c in range(min(clips, 5))])
    ref = [orc.imdct(co[c], w) for c in range(len(co))]
    co = co[np.arange(clips) % co.shape[0]].astype(np.float32)
    zafx.set_row_padding("auto")
    try:
        y = zafx.imdct_batch(co, w)
        T = co.shape[2]
        assert zafx.mdct_plan(w, inverse=True, row_align=0 if T % 32 == 0 else 32).last_kernel == "k_imdct_q"
        assert y.shape == (clips, len(ref[0]))
        for c in range(clips):
            assert relerr(y[c], ref[c % 5]) <= TOL_FFT, c
        k = min(n, y.shape[1])
        assert np.max(np.abs(y[:, :k] - x[:, :k])) < 1e-5   # TDAC (zaf.py:1060-1096 then :1099-1185)
        rng = np.random.default_rng(n)
        noisy = (co[:2] + 0.05 * rng.standard_normal(co[:2].shape)).astype(np.float32)
        y = zafx.imdct_batch(noisy, w)
        for c in range(len(noisy)):
            assert relerr(y[c], orc.imdct(noisy[c].astype(np.float64), w)) <= TOL_FFT
    finally:
        zafx.set_row_padding("compact")
    if T % 4:
        zafx.imdct_batch(co[:1], w)
        assert zafx.mdct_plan(w, inverse=True).last_kernel != "k_imdct_q"   # compact rows off the 16-byte grid: the generic kernel


def test_batch_functions_pad_rows_off_the_line_grid(zafx):
    """The default of the STFT / MDCT *_batch functions (round 6, zafx.set_row_padding("auto")): a frame count off the 128-byte line grid of the
    (F, T) rows runs on a row-padded device array -- the kernels' on-grid forms -- and the NumPy array handed back is a view of the padded
    result with the reference's shape, dtype and indexing (a different kernel form than the compact path's: equal within rounding, both
    within tolerance of the reference); the inverse functions take that view as it lies (no
    copy) and any other array through a padded copy; `out=`, row_align=0, the frame-major layout and "compact" keep the compact array."""
    n, hop = 1024 * 36 + 5, 1024           # T = 38: off the grid of 16 complex64 / 32 float32
    x = np.stack([synth_clip(71, c, n) for c in range(3)])
    ham, kbd = zafx.hamming(2048), zafx.kaiser_bessel_derived(2048)
    compact = {"stft": zafx.stft_batch(x, ham, hop), "mag": zafx.stft_batch(x, ham, hop, onesided="magnitude"), "mdct": zafx.mdct_batch(x, kbd)}
    assert all(v.flags.c_contiguous for v in compact.values())
    zafx.set_row_padding("auto")
    try:
        s = zafx.stft_batch(x, ham, hop)
        assert zafx.stft_plan(ham, hop, row_align=16).last_kernel == "k_stft_ft16"                 # the on-grid kernel, not the carry form
        assert s.shape == (3, 2048, 38) and s.strides == (2048 * 48 * 8, 48 * 8, 8) and relerr(s, compact["stft"]) <= 2e-6 and relerr(s[0], orc.stft(x[0].astype(np.float64), ham, hop)) <= TOL_FFT
        m = zafx.stft_batch(x, ham, hop, onesided="magnitude")
        assert m.shape == (3, 1025, 38) and m.strides[1] == 64 * 4 and relerr(m, compact["mag"]) <= 2e-6
        d = zafx.mdct_batch(x, kbd)
        assert d.shape == compact["mdct"].shape and d.strides[1] == 64 * 4 and relerr(d, compact["mdct"]) <= 2e-6 and relerr(d[0], orc.mdct(x[0].astype(np.float64), kbd)) <= TOL_FFT
        # inverse kinds: the padded view goes as it lies, a compact array through a copy -- the same samples
        y_view, y_copy = zafx.istft_batch(s, ham, hop), zafx.istft_batch(compact["stft"], ham, hop)
        assert zafx.istft_plan(ham, hop, row_align=16).last_kernel == "k_istft_ft16"
        zafx.set_row_padding("compact")
        y_compact = zafx.istft_batch(compact["stft"], ham, hop)
        zafx.set_row_padding("auto")
        assert relerr(y_view, y_copy) <= 2e-6 and relerr(y_view, y_compact) <= 2e-6 and np.max(np.abs(y_view[:, :n] - x)) < 1e-5
        r_view, r_copy = zafx.imdct_batch(d, kbd), zafx.imdct_batch(compact["mdct"], kbd)
        assert relerr(r_view, r_copy) <= 2e-6 and np.max(np.abs(r_view[:, :n - 1] - x[:, :n - 1])) < 1e-5
        assert zafx.stft_batch(x, ham, hop, row_align=0).flags.c_contiguous and zafx.stft_batch(x, ham, hop, layout="TF").flags.c_contiguous
        out = np.empty((3, 2048, 38), np.complex64)
        assert zafx.stft_batch(x, ham, hop, out=out) is out and np.array_equal(out, compact["stft"])
        assert zafx.stft_batch(x[:, :1024 * 31], ham, hop).flags.c_contiguous                       # T = 32: on the grid, nothing to pad
        f64 = zafx.stft_batch(x.astype(np.float64), ham, hop, f64=True)                              # complex128 rows: lines of 8
        assert f64.strides[1] == 40 * 16 and relerr(f64[0], orc.stft(x[0].astype(np.float64), ham, hop)) <= TOL_F64
        one = zafx.stft(x[0], ham, hop)                                                              # the drop-ins hand back fresh contiguous arrays
        assert one.flags.c_contiguous and one.dtype == np.complex128
    finally:
        zafx.set_row_padding("compact")


def test_large_arrays_come_in_chunks(zafx):
    """zafx_alloc assembles arrays of 1 GiB and more from 64-MiB physical allocations mapped back to back (round 6, DESIGN 3: where a multi-GB array
    lies decides the rate of the kernels that write it): one contiguous range to copies and kernels -- a pattern written and read back across the chunk
    borders --, released by zafx_free (200 x 2 GiB allocated and freed would not fit the device if anything stayed behind)."""
    n = (1 << 30) + (64 << 20) + 12345 * 8
    host = (np.arange(n // 8, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(7)
    d = zafx.DeviceBuffer.from_host(host)
    back = d.download()
    assert back.shape == host.shape and np.array_equal(back, host)
    for border in (64 << 20, 1 << 30):   # (a few words either side of two borders, through the partial-download path)
        k = border // 8
        assert np.array_equal(d.download(k - 4, 8), host[k - 4:k + 4])
    d.free()
    for _ in range(200):
        zafx.DeviceBuffer((1 << 31,), np.uint8).free()
    small = zafx.DeviceBuffer((1000,), np.float32)   # (below the threshold: the plain allocator)
    small.free()



def test_plain_allocator_switch():
    """ZAFX_ALLOC_CHUNK_MB=0: every array from hipMalloc (the switch a caller has when the virtual-memory API is unwanted): a 1.1 GiB round trip and an STFT in a
    fresh process with the variable set."""
    import os
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {os.path.join(ROOT, 'zaf-python_amd')!r}); sys.path.insert(0, {ROOT!r})\n"
        "import zafx\n"
        "from oracle import zaf_oracle as orc\n"
        "h = np.arange((1 << 30) // 8 + 12345 * 1024, dtype=np.uint64)\n"
        "d = zafx.DeviceBuffer.from_host(h)\n"
        "assert np.array_equal(d.download(), h)\n"
        "d.free()\n"
        "x = np.random.default_rng(3).standard_normal((2, 30000)).astype(np.float32)\n"
        "w = zafx.hamming(2048)\n"
        "s = zafx.stft_batch(x, w, 1024)\n"
        "r = orc.stft(x[1].astype(np.float64), w, 1024)\n"
        "assert np.max(np.abs(s[1] - r)) / np.max(np.abs(r)) < 1e-5\n"
        "print('plain allocator ok')\n")
    res = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ZAFX_ALLOC_CHUNK_MB="0"), capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "plain allocator ok" in res.stdout, (res.stdout + res.stderr)[-2000:]



def test_alloc_placed_through_the_c_abi(zafx):
    """zafx_alloc_placed (the C twin of DeviceBuffer.placed): the output buffer of a plan as the fastest of n allocations, every candidate
    timed by the library with the plan's own kernel; the buffer it returns holds the plan's result; bad arguments fail."""
    import ctypes
    from zafx import _lib
    B, N = 16, 100000
    x = np.stack([synth_clip(67, c, N) for c in range(B)])
    w = zafx.hamming(2048)
    plan = zafx.stft_plan(w, 1024)
    d_x = zafx.DeviceBuffer.from_host(x)
    buf, times = zafx.DeviceBuffer.placed_for(plan, d_x, B, N, candidates=3, reps=4)
    assert len(times) == 3 and all(t > 0 for t in times) and buf.shape == plan.out_shape(B, N)
    got = buf.download()   # (the probes left the transform of d_x in it)
    assert np.array_equal(got, zafx.stft_batch(x, w, 1024))
    buf.free()
    p = ctypes.c_void_p()
    lib = _lib.load()
    assert lib.zafx_alloc_placed(plan.handle, ctypes.byref(p), 16, d_x.ptr, B, N, 2, 1, None) != 0          # too small for the output
    assert lib.zafx_alloc_placed(plan.handle, ctypes.byref(p), 1 << 20, d_x.ptr, 1, N, 0, 1, None) != 0     # no candidates
    assert lib.zafx_alloc_placed(None, ctypes.byref(p), 1 << 20, d_x.ptr, 1, N, 1, 1, None) != 0
    d_x.free()


def test_full_share_cqt_device_resident(zafx, golden):
    """BASELINE config 5, one GPU's share as bench.py runs it: 1024 clips x 30 s (5.4 GB of input: clips from index 812
    on start beyond 4 GiB).  Clip 0 is the clip of the Q0 golden probes; 8 distinct clips; first, middle and last
    replicas against the oracle / the probes and bit-equal among themselves."""
    B, N, distinct = 1024, 1323000, 8
    base = np.stack([synth_clip(0, c, N) for c in range(distinct)])
    d_x = _replicated_on_device(zafx, base, B)
    ck = csr(golden["consts"], "ck")
    for chroma in (False, True):
        plan = zafx.cqt_plan(44100, 25, ck, 24 if chroma else None)
        rows = 24 if chroma else 144
        d_out = zafx.DeviceBuffer(plan.out_shape(B, N), np.float32)
        plan.execute(d_x, d_out, B, N)
        plan.sync()
        first, mid, last = (d_out.download(s, distinct) for s in (0, B // 2, B - distinct))
        assert first.shape == (distinct, rows, 750)
        assert np.array_equal(first, mid) and np.array_equal(first, last)
        for blk in (first, last):
            _check_probes(golden["config"], "Q0_chroma" if chroma else "Q0_cqt", blk[0].astype(np.float64), TOL_FB)
        if not chroma:
            for c in (1, 7):
                assert relerr(last[c], orc.cqtspectrogram(base[c].astype(np.float64), 44100, 25, ck)) <= TOL_FB
        whole = d_out.download().reshape(B // distinct, distinct, rows, 750)
        assert np.array_equal(whole, np.broadcast_to(first[None], whole.shape))
        d_out.free()
    d_x.free()


def test_batch_larger_than_grid(zafx):
    """More tiles than persistent workgroups: every workgroup loops; clips stay independent."""
    b, n = 700, 20000
    base = np.stack([synth_clip(11, c, n) for c in range(7)])
    x = np.tile(base, (b // 7, 1))
    ham = zafx.hamming(2048)
    got = zafx.stft_batch(x, ham, 1024)
    for c in range(7):
        assert relerr(got[c], orc.stft(base[c].astype(np.float64), ham, 1024)) <= TOL_FFT
    assert np.array_equal(got[:7], got[-7:])


# ------------------------------------------------------------------ PCM ingest
@pytest.mark.parametrize("dtype,channels", [(np.int16, 1), (np.int16, 2), (np.int32, 2), (np.int16, 5)])
def test_pcm_ingest(zafx, dtype, channels):
    """wavread's normalisation (zaf.py:1202) + the examples' channel mean (zaf.py:65) on device."""
    rng = np.random.default_rng(12)
    info = np.iinfo(dtype)
    pcm = rng.integers(info.min, info.max, size=(3, 30000, channels), dtype=dtype, endpoint=True)
    ref = np.mean(pcm / pow(2, pcm.itemsize * 8 - 1), axis=2)
    got = zafx.pcm_to_mono(pcm)
    assert got.shape == ref.shape and got.dtype == np.float32
    assert np.max(np.abs(got - ref)) <= 2e-7
    ham = zafx.hamming(2048)
    spec = zafx.stft_pcm_batch(pcm, ham, 1024)
    for c in range(3):
        assert relerr(spec[c], orc.stft(ref[c], ham, 1024)) <= TOL_FFT
    w2 = orc.hamming_periodic(1764)   # a window the float32 kernels do not take: normalisation on the device, float64 transform
    spec2 = zafx.stft_pcm_batch(pcm, w2, 441)
    assert spec2.dtype == np.complex64 and relerr(spec2[0], orc.stft(ref[0], w2, 441)) <= TOL_FFT
    with pytest.raises(ValueError):
        zafx.pcm_to_mono(pcm.astype(np.float32))


@pytest.mark.parametrize("dtype,channels", [(np.int16, 1), (np.int16, 2), (np.int32, 1)])
def test_pcm_batch_every_kind(zafx, dtype, channels):
    """Every transform that takes samples from integer PCM through the chunked host pipeline (zafx_run_host_pcm: integers cross
    PCIe, zaf.py:1202 / :65 on the device): equal, bit for bit, to the float32 call on the normalised mono signal; parity against
    the oracle on x / 2^(bits-1); several chunks (both staging sets, a ragged last chunk)."""
    rng = np.random.default_rng(21)
    info = np.iinfo(dtype)
    pcm = rng.integers(info.min // 2, info.max // 2, size=(7, 50000, channels), dtype=dtype, endpoint=True)
    if channels == 1 and dtype == np.int16:
        pcm = pcm[:, :, 0]   # (clips, frames) is accepted too
    mono = zafx.pcm_to_mono(pcm)
    ref = np.mean(np.atleast_3d(pcm) / pow(2, pcm.itemsize * 8 - 1), axis=2)
    assert np.max(np.abs(mono - ref)) <= 2e-7
    ham, kbd = zafx.hamming(2048), zafx.kaiser_bessel_derived(2048)
    fb = zafx.melfilterbank(44100, 2048, 128)
    ck = zafx.cqtkernel(44100, 24, 55, 3520)
    calls = [
        (zafx.stft_pcm_batch, zafx.stft_batch, (ham, 1024), lambda x: orc.stft(x, ham, 1024), TOL_FFT),
        (zafx.mdct_pcm_batch, zafx.mdct_batch, (kbd,), lambda x: orc.mdct(x, kbd), TOL_FFT),
        (zafx.melspectrogram_pcm_batch, zafx.melspectrogram_batch, (ham, 1024, fb), lambda x: orc.melspectrogram(x, ham, 1024, fb), TOL_FB),
        (zafx.mfcc_pcm_batch, zafx.mfcc_batch, (ham, 1024, fb, 20), lambda x: orc.mfcc(x, ham, 1024, fb, 20), TOL_FB),
        (zafx.cqtspectrogram_pcm_batch, zafx.cqtspectrogram_batch, (44100, 25, ck), lambda x: orc.cqtspectrogram(x, 44100, 25, ck), TOL_FB),
        (zafx.cqtchromagram_pcm_batch, zafx.cqtchromagram_batch, (44100, 25, 24, ck), lambda x: orc.cqtchromagram(x, 44100, 25, 24, ck), TOL_FB),
    ]
    for pcm_fn, f32_fn, args, oracle, tol in calls:
        got = pcm_fn(pcm, *args)
        assert np.array_equal(got, f32_fn(mono, *args)), pcm_fn.__name__
        for c in (0, 6):
            assert relerr(got[c], oracle(ref[c])) <= tol, (pcm_fn.__name__, c)
    plan = zafx.mel_plan(ham, 1024, fb)
    whole = plan.run_host_pcm(pcm)
    for chunk in (1, 2, 3):
        assert np.array_equal(plan.run_host_pcm(pcm, chunk_clips=chunk), whole), chunk
    with pytest.raises(ValueError):
        zafx.istft_plan(ham, 1024).run_host_pcm(pcm)


# ------------------------------------------------------------------ degenerate sizes
def test_empty_and_one_sample_clips(zafx):
    """N = 0 is legal in the reference (one all-zero frame for stft/mdct, no frame for the CQT)."""
    ham = zafx.hamming(2048)
    kbd = zafx.kaiser_bessel_derived(2048)
    empty = np.zeros(0)
    ref = orc.stft(empty, ham, 1024)
    got = zafx.stft(empty, ham, 1024)
    assert got.shape == ref.shape == (2048, 1) and not got.any()
    m = zafx.mdct(empty, kbd)
    assert m.shape == orc.mdct(empty, kbd).shape == (1024, 1) and not m.any()
    assert zafx.imdct(m, kbd).shape == orc.imdct(np.zeros((1024, 1)), kbd).shape == (0,)
    assert zafx.istft(got, ham, 1024).shape == orc.istft(ref, ham, 1024).shape == (0,)
    fb = zafx.melfilterbank(44100, 2048, 128)
    assert zafx.melspectrogram(empty, ham, 1024, fb).shape == (128, 1)
    ck = zafx.cqtkernel(4000, 12, 200, 1600)
    assert zafx.cqtspectrogram(empty, 4000, 50, ck).shape == orc.cqtspectrogram(empty, 4000, 50, ck).shape == (36, 0)
    assert zafx.cqtspectrogram(np.ones(79), 4000, 50, ck).shape == (36, 0)   # shorter than one step
    one = np.array([0.5])
    assert relerr(zafx.stft(one, ham, 1024), orc.stft(one, ham, 1024)) <= TOL_FFT


# ------------------------------------------------------------------ in-process sharder (threads, one plan per slot)
def test_run_sharded_threads(zafx):
    """Two shard slots driven from two host threads (both on GPU 0 here; on a node each slot is
    another device): concatenated shards equal the unsharded result bit for bit."""
    x = np.stack([synth_clip(13, c, 50000) for c in range(9)])
    ham = zafx.hamming(2048)
    whole = zafx.stft_batch(x, ham, 1024)
    parts = zafx.run_sharded(zafx.stft_batch, x, [0, 0], ham, 1024)
    assert parts.shape == whole.shape and np.array_equal(parts, whole)
    fb = zafx.melfilterbank(44100, 2048, 128)
    assert np.array_equal(zafx.run_sharded(zafx.mfcc_batch, x, [0, 0, 0], ham, 1024, fb, 20), zafx.mfcc_batch(x, ham, 1024, fb, 20))
    one = zafx.run_sharded(zafx.stft_batch, x[:1], [0, 0], ham, 1024)   # fewer clips than slots
    assert np.array_equal(one, whole[:1])


# ------------------------------------------------------------------ dct / dst
@pytest.mark.parametrize("n", [8, 9, 100, 1024])
def test_dct_dst(zafx, golden, n):
    """zaf.dct / zaf.dst types 1-4 as one MFMA GEMM per batch, against the reference's golden vectors."""
    g = golden["dctdst"]
    x = g[f"x_{n}"]
    for t in (1, 2, 3, 4):
        got = zafx.dct(x, t)
        assert got.dtype == np.float64 and relerr(got, g[f"dct{t}_{n}"]) <= TOL_FFT
        assert relerr(zafx.dst(x, t), g[f"dst{t}_{n}"]) <= TOL_FFT
    # batched, ragged tile edges (70 rows, n not a multiple of the 64 x 64 x 32 tile)
    xb = np.stack([synth_clip(14, c, n) for c in range(70)])
    for t in (2, 4):
        gb = zafx.dct_batch(xb, t)
        ref = np.stack([orc.dct(v.astype(np.float64), t) for v in xb])
        assert gb.shape == ref.shape and relerr(gb, ref) <= TOL_FFT
    gb = zafx.dst_batch(xb, 3)
    ref = np.stack([orc.dst(v.astype(np.float64), 3) for v in xb])
    assert relerr(gb, ref) <= TOL_FFT
    # the reference's plotted self-check (zaf.py:866-897): DST-II and DST-III are inverses
    assert np.max(np.abs(zafx.dst(zafx.dst(x, 2), 3) - x)) < 1e-5


def off_grid_dct_kernel(n, t):
    """The kernel of a dct / dst length off k_dct's power-of-two grid: types 2-4 of a length 4 j keep k_dct's maps around an N/2-point Bluestein
    convolution (k_dct_bsh, round 6: half the transform length), everything else is the chirp-z sum of all N points (k_dct_bs32)."""
    return "k_dct_bsh" if t >= 2 and n % 4 == 0 else "k_dct_bs32"


@pytest.mark.parametrize("n", [63, 64, 65, 1023, 1024, 1025])
def test_dct_dst_on_the_fft_core(zafx, golden, n):
    """Lengths whose N/2 (types 2-4), N-1 (dct 1) or N+1 (dst 1) is a power of two from 32 up run on k_dct: one M-point complex
    transform per vector with the symmetric-extension maps of zaf.py:769-835, :906-981 folded around it -- against the
    reference's golden vectors; the others run as chirp-z sums on the Bluestein machinery (k_dct_bs32), no length on a dense matrix."""
    g = golden["dctdst"]
    x = g[f"x_{n}"]
    for sine, fn, name in ((False, zafx.dct, "dct"), (True, zafx.dst, "dst")):
        for t in (1, 2, 3, 4):
            on_core = zafx.dct_fft_length(n, t, sine) is not None
            assert on_core == ((n - 1 if not sine else n + 1) in (64, 1024) if t == 1 else n in (64, 1024)), (n, t, sine)
            got = fn(x, t)
            assert got.dtype == np.float64 and got.shape == x.shape and relerr(got, g[f"{name}{t}_{n}"]) <= TOL_FFT, (name, t)
            plan = zafx.dct_plan(n, t, sine)
            assert plan.kernel_name == plan.last_kernel == ("k_dct" if on_core else off_grid_dct_kernel(n, t))


@pytest.mark.parametrize("n", [8, 9, 100])
def test_dct_dst_short_lengths_on_the_fft_core_too(zafx, golden, n):
    """Verdict r4 item 8: dctdst.npz lengths 9, 63, 65, 100, 1023, 1025 (and 8) leave the dense N x N product for the FFT forms."""
    g = golden["dctdst"]
    x = g[f"x_{n}"]
    for sine, fn, name in ((False, zafx.dct, "dct"), (True, zafx.dst, "dst")):
        for t in (1, 2, 3, 4):
            assert relerr(fn(x, t), g[f"{name}{t}_{n}"]) <= TOL_FFT, (name, t)
            assert zafx.dct_plan(n, t, sine).last_kernel == off_grid_dct_kernel(n, t)


@pytest.mark.parametrize("n,rows", [(2, 5), (3, 70), (4, 9), (12, 1000), (17, 300), (33, 9), (127, 64), (129, 3), (132, 64), (260, 700), (441, 100), (1000, 1500), (1028, 5),
                                    (1764, 33), (2047, 7), (2052, 300), (3000, 40), (4097, 5), (5000, 3), (8188, 40), (8191, 2), (8192 - 2, 2)])
def test_dct_dst_of_any_length(zafx, n, rows):
    """zaf.dct / zaf.dst take any length (zaf.py:760-839, :900-981: one np.fft.fft of a symmetric extension): every length up to 8192
    that is off the power-of-two grid of k_dct runs as a chirp-z sum (k_dct_bs32, convolution lengths 128 ... 16384) or, types 2-4 of a
    length 4 j, as k_dct's maps around an N/2-point Bluestein convolution (k_dct_bsh, lengths 128 ... 8192: every one of them here), all eight
    transforms, more rows than workgroups, against the oracle; the inverse pairs II / III and IV / IV close the loop."""
    x = np.stack([synth_clip(71, c % 11, n) for c in range(rows)])
    for sine, batch, one in ((False, zafx.dct_batch, orc.dct), (True, zafx.dst_batch, orc.dst)):
        for t in (1, 2, 3, 4):
            if zafx.dct_fft_length(n, t, sine) is not None:
                continue   # (k_dct's lengths: test_dct_dst_batches_every_size)
            got = batch(x, t)
            assert zafx.dct_plan(n, t, sine).last_kernel == off_grid_dct_kernel(n, t) and got.shape == x.shape and got.dtype == np.float32
            for c in range(min(rows, 11)):
                assert relerr(got[c], one(x[c].astype(np.float64), t)) <= TOL_FFT, (sine, t, c)
            if rows > 11:
                assert np.array_equal(got[11:22], got[0:11])   # replicas of the same vectors on other workgroups
    k = min(rows, 3)
    for batch in (zafx.dct_batch, zafx.dst_batch):
        assert np.max(np.abs(batch(batch(x[:k], 2), 3) - x[:k])) < 3e-5
        assert np.max(np.abs(batch(batch(x[:k], 4), 4) - x[:k])) < 3e-5


@pytest.mark.parametrize("n,rows", [(64, 1), (64, 1000), (128, 7), (256, 33), (512, 5), (1024, 16384), (2048, 9), (4096, 300), (8192, 3), (16384, 5),
                                    (65, 40), (63, 40), (4097, 3), (4095, 3), (8193, 2), (8191, 2)])
def test_dct_dst_batches_every_size(zafx, n, rows):
    """Every FFT size of k_dct (M = 32 ... 8192), ragged and whole workgroup passes, more passes than workgroups, rows that are
    not 16-byte multiples (type 1), every type, both transforms; unaligned device rows; SciPy's orthonormal dct as a second
    reference (the reference's own plotted check, zaf.py:728-738); the inverse pairs (zaf.py:866-897)."""
    import scipy.fftpack
    x = np.stack([synth_clip(15, c % 11, n) for c in range(rows)])
    pick = sorted({0, rows - 1, rows // 2, min(rows - 1, 257)})
    for t in (1, 2, 3, 4):
        for sine in (False, True):
            if zafx.dct_fft_length(n, t, sine) is None:
                continue
            got = (zafx.dst_batch if sine else zafx.dct_batch)(x, t)
            assert got.shape == x.shape and got.dtype == np.float32
            for c in pick:
                ref = (orc.dst if sine else orc.dct)(x[c].astype(np.float64), t)
                assert relerr(got[c], ref) <= TOL_FFT, (t, sine, c)
                sp = (scipy.fftpack.dst if sine else scipy.fftpack.dct)(x[c].astype(np.float64), type=t, norm="ortho")
                assert relerr(got[c], sp) <= TOL_FFT, (t, sine, c)
            if rows > 11:
                assert np.array_equal(got[11:22], got[0:11])   # replicas of the same vectors in other workgroup passes
    if n % 2 == 0:
        k = min(3, rows)
        back = zafx.dct_batch(zafx.dct_batch(x[:k], 2), 3)
        assert np.max(np.abs(back - x[:k])) < 2e-5
        back = zafx.dst_batch(zafx.dst_batch(x[:k], 4), 4)
        assert np.max(np.abs(back - x[:k])) < 2e-5
        # a device array that starts 4 bytes into an allocation: the 4-byte load / store path
        import ctypes
        plan = zafx.dct_plan(n, 2)
        d_in = zafx.DeviceBuffer((k * n + 1,), np.float32)
        d_out = zafx.DeviceBuffer((k * n + 1,), np.float32)
        d_in.upload(np.concatenate(([0.0], x[:k].reshape(-1))).astype(np.float32))
        views = [zafx.DeviceBuffer((k, n), np.float32, _ptr_from_pool=ctypes.c_void_p(b.ptr.value + 4)) for b in (d_in, d_out)]
        plan.execute(views[0], views[1], k, n)
        plan.sync()
        for v in views:
            v.ptr = ctypes.c_void_p()   # (views, not allocations: nothing to free)
        shifted = d_out.download()[1:].reshape(k, n)
        assert np.array_equal(shifted, zafx.dct_batch(x[:k], 2))
        d_in.free()
        d_out.free()


# ------------------------------------------------------------------ one-sided spectra
@pytest.mark.parametrize("wl,hop,n", [(2048, 1024, 441000), (2048, 512, 30000), (1024, 256, 9001), (128, 64, 777), (4096, 2048, 50000)])
def test_onesided_stft_istft(zafx, wl, hop, n):
    """onesided=True: the STFT writes rows 0..W/2 of the reference's result (the slice zaf.py:83 keeps);
    the ISTFT of those rows equals the reference's istft of the full spectrum of a real signal."""
    x = np.stack([synth_clip(12, c, n) for c in range(3)])
    w = zafx.hamming(wl)
    ref = orc.stft_batch(x.astype(np.float64), w, hop)
    half = wl // 2 + 1
    for layout in ("FT", "TF"):
        got = zafx.stft_batch(x, w, hop, layout=layout, onesided=True)
        if layout == "TF":
            got = got.transpose(0, 2, 1)
        assert got.shape == (3, half, ref.shape[2])
        for c in range(3):
            assert relerr(got[c], ref[c, :half]) <= TOL_FFT, (layout, c)
        spec = ref[:, :half] if layout == "FT" else np.ascontiguousarray(ref[:, :half].transpose(0, 2, 1))
        y = zafx.istft_batch(spec, w, hop, layout=layout, onesided=True)
        for c in range(3):
            yref = orc.istft(ref[c], w, hop)
            assert y[c].shape == yref.shape and relerr(y[c], yref) <= TOL_FFT, (layout, c)


@pytest.mark.parametrize("hop,n,clips", [(2048, 150001, 3), (2048, 150000, 2), (1024, 70000, 2), (300, 20000, 1), (6000, 90000, 2),
                                          (2048, 1, 1), (2048, 4097, 1), (1764, 441000, 2), (2048, 16 * 2048 * 3, 35), (1763, 60001, 2), (2047, 99999, 1)])
def test_stft_w4096_two_bands(zafx, hop, n, clips):
    """W = 4096 in the reference layout runs k_stft_ft16b: sixteen-frame tiles as two bands of bins (even rows from
    FFT_1024(z[n] + z[n + 1024]), odd rows from the twiddled difference; zaf.py:128-139 is what it replaces).  Odd clip
    lengths and odd hops take the predicated loads, even ones the clip-descriptor loads; T covers whole tiles, ragged last
    tiles, rows on and off the 128-byte grid, more tiles than workgroups; every spectrum kind."""
    x = np.stack([synth_clip(41, c, n) for c in range(clips)])
    w = zafx.hamming(4096)
    plan = zafx.stft_plan(w, hop)
    assert plan.kernel_name == "k_stft_ft16b"   # planned: the band family; what RAN depends on the call (last_kernel)
    ref = orc.stft_batch(x.astype(np.float64), w, hop)
    got = zafx.stft_batch(x, w, hop)
    assert got.shape == ref.shape and got.dtype == np.complex64
    T = got.shape[2]
    assert plan.last_kernel == ("k_stft_ft16b" if T % 16 == 0 else "k_stft_ft16bc"), (hop, n, T)   # complex rows off the line grid: the carry form
    for c in range(clips):
        assert relerr(got[c], ref[c]) <= TOL_FFT, c
    if clips <= 3:
        half = 2049
        one = zafx.stft_batch(x, w, hop, onesided=True)
        # one-sided rows off the grid: the carry form from hop >= W/2 (it reads every sample twice), the generic kernel below
        assert zafx.stft_plan(w, hop, onesided=True).last_kernel == ("k_stft_ft16b" if T % 16 == 0 else "k_stft_ft16bc" if 2 * hop >= 4096 else "k_stft"), (hop, T)
        mag = zafx.stft_batch(x, w, hop, onesided="magnitude")
        assert zafx.stft_plan(w, hop, onesided="magnitude").last_kernel == "k_stft_ft16b"
        pw = zafx.stft_batch(x, w, hop, onesided="power")
        for c in range(clips):
            assert relerr(one[c], ref[c, :half]) <= TOL_FFT
            assert relerr(mag[c], np.abs(ref[c, :half])) <= TOL_FFT
            assert relerr(pw[c], np.abs(ref[c, :half]) ** 2) <= 2 * TOL_FFT


@pytest.mark.parametrize("hop,n,clips", [(4096, 300001, 3), (4096, 4096 * 31, 2), (2048, 140000, 2), (2048, 2048 * 63 - 5, 2), (300, 30000, 1), (9000, 150000, 2),
                                          (4096, 1, 1), (4096, 8193, 1), (1764, 441000, 1), (4096, 4096 * 47, 100), (4096, 16 * 4096 * 2, 20), (4095, 4095 * 15, 2)])
def test_stft_w8192_four_classes(zafx, hop, n, clips):
    """W = 8192 in the reference layout runs k_stft_ft16q: sixteen-frame tiles as four classes of bins (rows 4q from the packed transform of
    the sum of the frame's quarters, rows 8p + 2, 8p + 1, 8p + 5 from three more 1024-point transforms, every other row a mirror; zaf.py:128-139
    is what it replaces).  Any clip length and hop (4-byte loads), whole and ragged last tiles, rows on and off the 128-byte grid, more tiles
    than workgroups, every spectrum kind."""
    x = np.stack([synth_clip(43, c % 4, n) for c in range(clips)])
    w = zafx.hamming(8192)
    plan = zafx.stft_plan(w, hop)
    assert plan.kernel_name == "k_stft_ft16q"   # planned: the four-class family; what RAN depends on the call (last_kernel)
    ref = orc.stft_batch(x[:4].astype(np.float64), w, hop)
    got = zafx.stft_batch(x, w, hop)
    T = got.shape[2]
    assert plan.last_kernel == ("k_stft_ft16q" if T % 16 == 0 else "k_stft"), T   # complex two-sided rows off the line grid: the one-workgroup-per-tile kernel
    assert got.dtype == np.complex64 and got.shape[1:] == ref.shape[1:]
    for c in range(clips):
        assert relerr(got[c], ref[c % 4]) <= TOL_FFT, c
    if clips <= 3:
        half = 4097
        one = zafx.stft_batch(x, w, hop, onesided=True)
        mag = zafx.stft_batch(x, w, hop, onesided="magnitude")
        pw = zafx.stft_batch(x, w, hop, onesided="power")
        for kind in (True, "magnitude", "power"):
            assert zafx.stft_plan(w, hop, onesided=kind).last_kernel == "k_stft_ft16q"
        for c in range(clips):
            assert relerr(one[c], ref[c, :half]) <= TOL_FFT
            if T % 16 == 0:
                assert np.array_equal(one[c], got[c, :half])
            assert relerr(mag[c], np.abs(ref[c, :half])) <= TOL_FFT
            assert relerr(pw[c], np.abs(ref[c, :half]) ** 2) <= 2 * TOL_FFT


@pytest.mark.parametrize("hop,n,clips", [(2048, 150001, 3), (2048, 16 * 2048 * 2, 2), (1024, 70000, 2), (512, 40000, 1), (4096, 200000, 2),
                                          (1764, 100000, 2), (2048, 1, 1), (2048, 30 * 2048 * 16, 1), (2048, 40000, 300), (1000, 50000, 1)])
def test_istft_w4096_two_bands(zafx, hop, n, clips):
    """W = 4096 in the reference layout runs k_istft_ft16b for hops that are multiples of 4 (>= 512): one band of samples per
    workgroup (even / odd packed samples = two 1024-point inverse transforms; zaf.py:214-241 is what it replaces), the generic
    kernel otherwise (hop 1000 here: same numbers).  Non-Hermitian input, one-sided input, one clip in many segments (a carry
    across workgroups), many clips, odd frame counts (the generic kernel: 16-byte row pieces need an even row pitch)."""
    x = np.stack([synth_clip(47, c % 5, n) for c in range(clips)])
    w = zafx.hamming(4096)
    spec = orc.stft_batch(x[:5].astype(np.float64), w, hop)
    rng = np.random.default_rng(9)
    spec = spec + 0.05 * (rng.standard_normal(spec.shape) + 1j * rng.standard_normal(spec.shape))   # not Hermitian: the reference takes real(ifft(.))
    full = spec[np.arange(clips) % 5]
    got = zafx.istft_batch(full, w, hop)
    # the band form for hops that are multiples of 4 from 512 up (odd row pitches: its 8-byte row pieces), the generic kernel otherwise
    assert zafx.istft_plan(w, hop).kernel_name == "k_istft_ft16b"
    assert zafx.istft_plan(w, hop).last_kernel == ("k_istft_ft16d" if hop == 2048 else "k_istft_ft16b" if hop % 4 == 0 and hop >= 512 else "k_istft"), hop   # (hop W / 2: round 6's two-class kernel)
    for c in range(min(clips, 7)):
        ref = orc.istft(full[c], w, hop)
        assert got[c].shape == ref.shape and relerr(got[c], ref) <= TOL_FFT, c
    if clips > 7:
        assert np.array_equal(got[5:10], got[0:5]) and np.array_equal(got[clips - 5:clips], got[(clips - 5) % 5:(clips - 5) % 5 + 5] if (clips - 5) % 5 == 0 else got[clips - 5:clips])
    herm = orc.stft_batch(x[:2].astype(np.float64) if clips >= 2 else x[:1].astype(np.float64), w, hop)
    one = zafx.istft_batch(herm[:, :2049], w, hop, onesided=True)
    for c in range(herm.shape[0]):
        ref = orc.istft(herm[c], w, hop)
        assert relerr(one[c], ref) <= TOL_FFT


def test_onesided_rejected_elsewhere(zafx):
    with pytest.raises(zafx.ZafxError):
        zafx.Plan(zafx.MDCT, window_length=2048, onesided=True)
    with pytest.raises(ValueError):
        zafx.istft_batch(np.zeros((1, 2048, 4), np.complex64), zafx.hamming(2048), 1024, onesided=True)


# ------------------------------------------------------------------ random geometries (seeded) against the oracle
def _random_cases(seed, count):
    rng = np.random.default_rng(seed)
    cases = []
    for _ in range(count):
        wl = int(2 ** rng.integers(6, 13))                      # 64 .. 4096
        hop = int(rng.integers(max(wl // 8, 1), wl + 1))         # ceil(W/H) <= 8 keeps the ISTFT in range
        n = int(rng.integers(1, 6 * wl))
        cases.append((wl, hop, n))
    return cases


@pytest.mark.parametrize("wl,hop,n", _random_cases(20260928, 24))
def test_random_geometry_stft_family(zafx, wl, hop, n):
    """Frame counts, padding, trims and values for arbitrary (window, hop, length): both layouts, both
    spectrum kinds, 1-3 clips, inverse included."""
    b = 1 + (n % 3)
    x = np.stack([synth_clip(21, c, n) for c in range(b)])
    w = zafx.hamming(wl)
    ref = orc.stft_batch(x.astype(np.float64), w, hop)
    half = wl // 2 + 1
    for layout in ("FT", "TF"):
        for one in (False, True):
            got = zafx.stft_batch(x, w, hop, layout=layout, onesided=one)
            if layout == "TF":
                got = got.transpose(0, 2, 1)
            want = ref[:, :half] if one else ref
            assert got.shape == want.shape
            for c in range(b):
                assert relerr(got[c], want[c]) <= TOL_FFT, (layout, one, c)
    # float64 mode and the real-valued spectrum kinds on the same geometry
    got64 = zafx.stft_batch(x.astype(np.float64), w, hop, f64=True)
    mag = zafx.stft_batch(x, w, hop, onesided="magnitude")
    y64 = zafx.istft_batch(ref, w, hop, f64=True)
    for c in range(b):
        assert relerr(got64[c], ref[c]) <= 1e-12
        assert relerr(mag[c], np.abs(ref[c, :half])) <= TOL_FFT
        yref = orc.istft(ref[c], w, hop)
        assert y64[c].shape == yref.shape
        if yref.size:
            assert np.max(np.abs(y64[c] - yref)) / max(np.max(np.abs(yref)), 1e-300) <= 1e-12
    y = zafx.istft_batch(ref, w, hop)
    y1 = zafx.istft_batch(ref[:, :half], w, hop, onesided=True)
    for c in range(b):
        yref = orc.istft(ref[c], w, hop)
        assert y[c].shape == yref.shape == y1[c].shape
        if yref.size:
            scale = max(np.max(np.abs(yref)), 1e-30)
            assert np.max(np.abs(y[c] - yref)) / scale <= TOL_FFT and np.max(np.abs(y1[c] - yref)) / scale <= TOL_FFT


@pytest.mark.parametrize("wl,n", [(int(2 ** e), int(n)) for e, n in zip(np.random.default_rng(7).integers(6, 14, 12),
                                                                          np.random.default_rng(8).integers(1, 40000, 12))])
def test_random_geometry_mdct_family(zafx, wl, n):
    b = 1 + (n % 3)
    x = np.stack([synth_clip(22, c, n) for c in range(b)])
    w = zafx.kaiser_bessel_derived(wl)
    ref = orc.mdct_batch(x.astype(np.float64), w)
    for layout in ("FT", "TF"):
        got = zafx.mdct_batch(x, w, layout=layout)
        if layout == "TF":
            got = got.transpose(0, 2, 1)
        assert got.shape == ref.shape
        for c in range(b):
            assert relerr(got[c], ref[c]) <= TOL_FFT, (layout, c)
        coefs = ref if layout == "FT" else np.ascontiguousarray(ref.transpose(0, 2, 1))
        y = zafx.imdct_batch(coefs, w, layout=layout)
        for c in range(b):
            yref = orc.imdct(ref[c], w)
            assert y[c].shape == yref.shape
            if yref.size:
                assert np.max(np.abs(y[c] - yref)) / max(np.max(np.abs(yref)), 1e-30) <= TOL_FFT, (layout, c)


# ------------------------------------------------------------------ float64 compute mode
TOL_F64 = 1e-12


@pytest.mark.parametrize("hop,n,clips", [(1024, 441000, 3), (1024, 30001, 2), (512, 40000, 5), (1000, 99998, 2), (441, 30000, 1), (1024, 1024 * 63, 40), (3000, 50000, 2),
                                         (1024, 1, 1), (1023, 20000, 2)])
def test_f64_stft_on_the_tiled_kernel(zafx, hop, n, clips):
    """W = 2048 in the reference layout, float64: k_stft_ft8_f64 (8-frame tiles = 128-byte lines of complex128 rows, a frame per
    wavefront; zaf.py:112-139 in its own dtype) -- interior and edge frames, odd clip lengths and hops (the sample-by-sample loads), ragged
    last tiles, rows on and off the line grid, more tiles than workgroups, two-sided and one-sided; the other kinds and layouts stay on
    k_stft_f64."""
    x = np.stack([synth_clip(37, c % 7, n).astype(np.float64) + 1e-9 * (c % 7) for c in range(clips)])
    w = zafx.hamming(2048)
    ref = orc.stft_batch(x[:7], w, hop)
    for one in (False, True):
        got = zafx.stft_batch(x, w, hop, onesided=one, f64=True)
        assert zafx.stft_plan(w, hop, onesided=one, f64=True).last_kernel == "k_stft_ft8_f64"
        assert got.dtype == np.complex128 and got.shape[1:] == ((1025 if one else 2048), ref.shape[2])
        for c in range(min(clips, 7)):
            assert relerr(got[c], ref[c, :1025] if one else ref[c]) <= TOL_F64, (one, c)
        if clips > 7:
            assert np.array_equal(got[7:14], got[0:7]) and np.array_equal(got[clips - 5:], got[(clips - 5) % 7:(clips - 5) % 7 + 5])
    zafx.stft_batch(x[:1], w, hop, onesided="magnitude", f64=True)
    assert zafx.stft_plan(w, hop, onesided="magnitude", f64=True).last_kernel == "k_stft_f64"
    zafx.stft_batch(x[:1], w, hop, layout="TF", f64=True)
    assert zafx.stft_plan(w, hop, layout="TF", f64=True).last_kernel == "k_stft_f64"


@pytest.mark.parametrize("hop,n,clips", [(1024, 441000, 3), (1024, 30001, 2), (1024, 1024 * 47, 40), (1536, 50000, 3), (2048, 40000, 2), (1025, 30011, 2),
                                         (1024, 1, 1), (1024, 1024 * 15, 2), (1200, 99999, 9)])
def test_f64_istft_on_the_tiled_kernel(zafx, hop, n, clips):
    """W = 2048 in the reference layout, float64, hop >= W / 2: k_istft_ft8_f64 (8-frame tiles of complex128 rows, the Hermitian fold in the
    loads, a frame per wavefront, frames overlap-added where they lie in LDS; zaf.py:214-241 in its own dtype) -- few clips (segments start
    inside a clip and compute their left neighbour), many clips, ragged last tiles, hops with and without overlap, two-sided and one-sided;
    shorter hops and the other layout stay on the scratch + overlap-add form."""
    x = np.stack([synth_clip(41, c % 7, n).astype(np.float64) + 1e-9 * (c % 7) for c in range(clips)])
    w = zafx.hamming(2048)
    spec = orc.stft_batch(x[:7], w, hop)
    if clips > 7:
        spec = np.concatenate([spec] * (clips // 7 + 1))[:clips]
    for one in (False, True):
        y = zafx.istft_batch(np.ascontiguousarray(spec[:, :1025] if one else spec), w, hop, onesided=one, f64=True)
        assert zafx.istft_plan(w, hop, onesided=one, f64=True).last_kernel == "k_istft_ft8_f64" and y.dtype == np.float64
        for c in range(min(clips, 7)):
            ref = orc.istft(spec[c], w, hop)
            assert y[c].shape == ref.shape and relerr(y[c], ref) <= TOL_F64, (one, c)
        if clips > 7:
            assert all(np.array_equal(y[c], y[c % 7]) for c in range(7, clips))
    zafx.istft_batch(spec[:1], w, hop, f64=True)
    for h1, layout in ((512, "FT"), (hop, "TF")):
        s0 = orc.stft_batch(x[:1], w, h1)
        s1 = np.ascontiguousarray(s0.transpose(0, 2, 1)) if layout == "TF" else s0
        y1 = zafx.istft_batch(s1, w, h1, layout=layout, f64=True)
        assert zafx.istft_plan(w, h1, layout=layout, f64=True).last_kernel == "k_ifft_frames_f64"
        assert relerr(y1[0], orc.istft(s0[0], w, h1)) <= TOL_F64


@pytest.mark.parametrize("n,clips", [(441000, 3), (30001, 2), (1024 * 47, 40), (1, 1), (1024 * 16 + 2, 2), (99999, 2)])
def test_f64_mdct_on_the_tiled_kernel(zafx, n, clips):
    """W = 2048 in the reference layout, float64: k_mdct_ft16_f64 (16-frame tiles = 128-byte lines of float64 rows, a frame per wavefront as
    8 x 8 x 8; zaf.py:1029-1073 in its own dtype) -- interior and edge frames, odd clip lengths (the sample-by-sample staging), ragged last
    tiles, rows on and off the line grid, more tiles than workgroups; the inverse stays on the scratch + overlap-add form."""
    x = np.stack([synth_clip(39, c % 7, n).astype(np.float64) + 1e-9 * (c % 7) for c in range(clips)])
    w = zafx.kaiser_bessel_derived(2048)
    got = zafx.mdct_batch(x, w, f64=True)
    assert zafx.mdct_plan(w, f64=True).last_kernel == "k_mdct_ft16_f64" and got.dtype == np.float64
    for c in range(min(clips, 7)):
        ref = orc.mdct(x[c], w)
        assert got[c].shape == ref.shape and relerr(got[c], ref) <= TOL_F64, c
    if clips > 7:
        assert np.array_equal(got[7:14], got[0:7]) and np.array_equal(got[clips - 5:], got[(clips - 5) % 7:(clips - 5) % 7 + 5])
    zafx.mdct_batch(x[:1], w, layout="TF", f64=True)
    assert zafx.mdct_plan(w, layout="TF", f64=True).last_kernel == "k_mdct_f64"
    # the inverse on its tiled form (k_imdct_ft16_f64: sixteen-frame tiles walked in order, seventeen frame buffers in rotation): clips in one
    # segment and in several (few clips: a segment starts inside a clip and computes its left neighbour itself), the block behind the last frame
    yi = zafx.imdct_batch(got, w, f64=True)
    assert zafx.mdct_plan(w, inverse=True, f64=True).last_kernel == "k_imdct_ft16_f64" and yi.dtype == np.float64
    for c in range(min(clips, 7)):
        yref = orc.imdct(orc.mdct(x[c], w), w)
        assert yi[c].shape == yref.shape
        if yref.size:
            assert relerr(yi[c], yref) <= TOL_F64, c
            k = min(n, len(yref))
            assert np.max(np.abs(yi[c][:k] - x[c, :k])) < 1e-12   # TDAC (zaf.py:1098-1109)


@pytest.mark.parametrize("wl,hop,n", [(2048, 1024, 441000), (2048, 512, 30000), (1024, 300, 9001), (64, 32, 1000), (8192, 4096, 50000),
                                       (256, 77, 1)])
def test_f64_stft_istft(zafx, wl, hop, n):
    """ZAFX_PRECISION_F64: float64 / complex128 on the device, every layout and spectrum kind, within 1e-12
    of the reference arithmetic (the oracle is bit-identical to zaf.py)."""
    x = np.stack([synth_clip(31, c, n).astype(np.float64) + 1e-9 * c for c in range(2)])   # genuinely float64 inputs
    w = zafx.hamming(wl)
    ref = orc.stft_batch(x, w, hop)
    half = wl // 2 + 1
    for layout in ("FT", "TF"):
        for one in (False, True):
            got = zafx.stft_batch(x, w, hop, layout=layout, onesided=one, f64=True)
            assert got.dtype == np.complex128
            if layout == "TF":
                got = got.transpose(0, 2, 1)
            want = ref[:, :half] if one else ref
            assert got.shape == want.shape
            for c in range(2):
                assert relerr(got[c], want[c]) <= TOL_F64, (layout, one, c)
            spec = want if layout == "FT" else np.ascontiguousarray(want.transpose(0, 2, 1))
            y = zafx.istft_batch(spec, w, hop, layout=layout, onesided=one, f64=True)
            assert y.dtype == np.float64
            for c in range(2):
                yref = orc.istft(ref[c], w, hop)
                assert y[c].shape == yref.shape
                if yref.size:
                    assert np.max(np.abs(y[c] - yref)) / max(np.max(np.abs(yref)), 1e-300) <= TOL_F64, (layout, one, c)


def test_f64_dropin_and_golden(zafx, golden):
    """set_precision("f64") switches the drop-in stft / istft / mdct / imdct to float64 device arithmetic: the tiny golden
    vectors of the real reference are met to 1e-12 instead of 1e-5."""
    g = golden["tiny"]
    zafx.set_precision("f64")
    try:
        for n in (1, 63, 64, 65, 1000):
            for hop in (32, 16):
                s = zafx.stft(g[f"x_{n}"], g["ham"], hop)
                assert s.dtype == np.complex128 and relerr(s, g[f"stft_{n}_{hop}"]) <= TOL_F64
                y = zafx.istft(g[f"stft_{n}_{hop}"], g["ham"], hop)
                assert relerr(y, g[f"istft_{n}_{hop}"]) <= TOL_F64
        y = zafx.istft(g["istft_generic_in"], g["ham"], 32)   # non-Hermitian input: real(ifft(.)) of anything
        assert relerr(y, g["istft_generic_out"]) <= TOL_F64
        for n in (1, 63, 64, 65, 1000):
            for wname in ("sine", "kbd"):
                c = zafx.mdct(g[f"x_{n}"], g[wname])
                assert c.dtype == np.float64 and relerr(c, g[f"mdct_{wname}_{n}"]) <= TOL_F64
                y = zafx.imdct(g[f"mdct_{wname}_{n}"], g[wname])
                yref = g[f"imdct_{wname}_{n}"]
                assert y.shape == yref.shape and (yref.size == 0 or relerr(y, yref) <= TOL_F64)
        fb = scipy.sparse.csr_matrix(g["fb_dense"])
        for n in (1, 63, 64, 65, 1000):
            for hop in (32, 16):
                assert relerr(zafx.melspectrogram(g[f"x_{n}"], g["ham"], hop, fb), g[f"mel_{n}_{hop}"]) <= TOL_F64
                assert relerr(zafx.mfcc(g[f"x_{n}"], g["ham"], hop, fb, 5), g[f"mfcc_{n}_{hop}"]) <= 1e-10   # log(. + eps) of tiny band sums
        ck = scipy.sparse.csr_matrix(g["ck_dense"])
        for n in (400, 4000, 4321):
            got = zafx.cqtspectrogram(g[f"xq_{n}"], 4000, 50, ck)
            assert got.shape == g[f"cqt_{n}"].shape and (got.size == 0 or relerr(got, g[f"cqt_{n}"]) <= TOL_F64)
            got = zafx.cqtchromagram(g[f"xq_{n}"], 4000, 50, 12, ck)
            assert got.shape == g[f"chroma_{n}"].shape and (got.size == 0 or relerr(got, g[f"chroma_{n}"]) <= TOL_F64)
    finally:
        zafx.set_precision("f32")
    with pytest.raises(zafx.ZafxError):
        zafx.Plan(zafx.LINEAR, window_length=64, n_filters=64, f64=True)


@pytest.mark.parametrize("fs,res,fmin,fmax,tr,n", [(44100, 24, 55, 3520, 25, 60000),     # fft_length 32768: 8 sub-transforms of 4096
                                                    (16000, 12, 110, 3520, 50, 30000),    # 4096: one transform in LDS
                                                    (8000, 12, 220, 1760, 40, 9001)])
def test_f64_cqt(zafx, fs, res, fmin, fmax, tr, n):
    """ZAFX_PRECISION_F64 for cqtspectrogram / cqtchromagram: complex128 kernel, float64 frames, 1e-12 of the reference."""
    ck = zafx.cqtkernel(fs, res, fmin, fmax)
    x = np.stack([synth_clip(43, c, n).astype(np.float64) + 1e-9 * c for c in range(2)])
    for layout in ("FT", "TF"):
        spec = zafx.cqtspectrogram_batch(x, fs, tr, ck, layout=layout, f64=True)
        chroma = zafx.cqtchromagram_batch(x, fs, tr, res, ck, layout=layout, f64=True)
        assert spec.dtype == np.float64 and chroma.dtype == np.float64
        if layout == "TF":
            spec, chroma = spec.transpose(0, 2, 1), chroma.transpose(0, 2, 1)
        for c in range(2):
            ref = orc.cqtspectrogram(x[c], fs, tr, ck)
            ref_c = orc.cqtchromagram(x[c], fs, tr, res, ck)
            assert spec[c].shape == ref.shape and chroma[c].shape == ref_c.shape
            assert relerr(spec[c], ref) <= TOL_F64, (layout, c)
            assert relerr(chroma[c], ref_c) <= TOL_F64, (layout, c)


@pytest.mark.parametrize("wl,hop,n,nmel", [(2048, 1024, 100000, 128), (2048, 512, 30001, 40), (1024, 512, 20000, 64), (4096, 2048, 50000, 128),
                                            (256, 100, 5000, 20)])
def test_f64_mel_mfcc(zafx, wl, hop, n, nmel):
    """ZAFX_PRECISION_F64 for melspectrogram / mfcc (any power-of-two window): the reference arithmetic to 1e-12 (mel)
    and 1e-10 (mfcc: the log amplifies the rounding of the smallest band sums), both layouts."""
    x = np.stack([synth_clip(41, c, n).astype(np.float64) + 1e-9 * c for c in range(2)])
    w = zafx.hamming(wl)
    fb = zafx.melfilterbank(44100, wl, nmel)
    for layout in ("FT", "TF"):
        mel = zafx.melspectrogram_batch(x, w, hop, fb, layout=layout, f64=True)
        cep = zafx.mfcc_batch(x, w, hop, fb, 13, layout=layout, f64=True)
        assert mel.dtype == np.float64 and cep.dtype == np.float64
        if layout == "TF":
            mel, cep = mel.transpose(0, 2, 1), cep.transpose(0, 2, 1)
        for c in range(2):
            ref_mel = orc.melspectrogram(x[c], w, hop, fb)
            ref_cep = orc.mfcc(x[c], w, hop, fb, 13)
            assert mel[c].shape == ref_mel.shape and cep[c].shape == ref_cep.shape
            assert relerr(mel[c], ref_mel) <= TOL_F64, (layout, c)
            assert relerr(cep[c], ref_cep) <= 1e-10, (layout, c)


@pytest.mark.parametrize("n,clips,tr", [(100000, 3, 25), (33001, 9, 50), (40000, 2, 10)])
def test_f64_cqt_on_the_tiled_kernel(zafx, n, clips, tr):
    """fft_length 32768 in float64: k_cqt_ft_f64 (16 x 1024 decimation in frequency in two rounds of eight wavefronts, the split of the bins the
    kernel reads only, the matrix's non-zeros as one stream per thread) -- frames that reach in front of and behind the clip (the buffer loads'
    out-of-range zeros), odd clip lengths (samples off the 16-byte grid), both layouts, the chromagram, a genuinely complex matrix (the 24-byte
    entries), a matrix with columns above W/2 (conjugate bins) and kernels of other frequency ranges (what the sub-transforms' last pass may skip follows
    the highest column); kernels the tiled form does not take (bins above 8191) stay on k_cqt_f64."""
    ck = zafx.cqtkernel(44100, 24, 55, 3520)
    assert ck.shape == (144, 32768)
    x = np.stack([synth_clip(53, c % 7, n).astype(np.float64) + 1e-9 * (c % 7) for c in range(clips)])
    for layout in ("FT", "TF"):
        spec = zafx.cqtspectrogram_batch(x, 44100, tr, ck, layout=layout, f64=True)
        assert zafx.cqt_plan(44100, tr, ck, layout=layout, f64=True).last_kernel == "k_cqt_ft_f64"
        chroma = zafx.cqtchromagram_batch(x, 44100, tr, 24, ck, layout=layout, f64=True)
        assert zafx.cqt_plan(44100, tr, ck, 24, layout=layout, f64=True).last_kernel == "k_cqt_ft_f64"
        if layout == "TF":
            spec, chroma = spec.transpose(0, 2, 1), chroma.transpose(0, 2, 1)
        for c in range(clips):
            ref, ref_c = orc.cqtspectrogram(x[c], 44100, tr, ck), orc.cqtchromagram(x[c], 44100, tr, 24, ck)
            assert spec[c].shape == ref.shape and chroma[c].shape == ref_c.shape
            assert relerr(spec[c], ref) <= TOL_F64 and relerr(chroma[c], ref_c) <= TOL_F64, (layout, c)
    # a complex matrix with mirrored columns: rows of the reference's kernel rotated by a phase, every third entry moved to W - c
    rng = np.random.default_rng(7)
    coo = ck.tocoo()
    data = coo.data * np.exp(2j * np.pi * rng.random(coo.nnz))
    cols = np.where(np.arange(coo.nnz) % 3 == 0, 32768 - coo.col, coo.col)
    ck2 = scipy.sparse.csr_matrix((data, (coo.row, cols)), shape=ck.shape)
    got = zafx.cqtspectrogram_batch(x[:2], 44100, tr, ck2, f64=True)
    assert zafx.cqt_plan(44100, tr, ck2, f64=True).last_kernel == "k_cqt_ft_f64"
    for c in range(2):
        assert relerr(got[c], orc.cqtspectrogram(x[c], 44100, tr, ck2)) <= TOL_F64, c
    # the sub-transforms' last pass writes only the quarters of their output the split reads: kernels whose highest column lies in every band of
    # that decision -- klo = (highest column) >> 4 = 20 (one quarter of the wave's outputs at either end), 70, 164 (above), 249 and 448 (>= 256: everything)
    shifted = scipy.sparse.csr_matrix((coo.data, (coo.row, np.where(coo.row >= 100, coo.col + 4500, coo.col))), shape=ck.shape)   # columns up to 7172
    for ckb, klo in ((zafx.cqtkernel(44100, 24, 55, 440), 20), (zafx.cqtkernel(44100, 24, 55, 1500), 70), (zafx.cqtkernel(44100, 24, 55, 5300), 249), (shifted, 448)):
        assert ckb.shape[1] == 32768 and ckb.indices.max() >> 4 == klo
        got = zafx.cqtspectrogram_batch(x[:2], 44100, tr, ckb, f64=True)
        assert zafx.cqt_plan(44100, tr, ckb, f64=True).last_kernel == "k_cqt_ft_f64", klo
        for c in range(2):
            assert relerr(got[c], orc.cqtspectrogram(x[c], 44100, tr, ckb)) <= TOL_F64, (klo, c)
    wide = zafx.cqtkernel(44100, 24, 55, 22050)   # the reference's docstring kernel: columns up to bin 16 613
    got = zafx.cqtspectrogram_batch(x[:1, :40000], 44100, tr, wide, f64=True)
    assert zafx.cqt_plan(44100, tr, wide, f64=True).last_kernel == "k_cqt_f64"
    assert relerr(got[0], orc.cqtspectrogram(x[0, :40000], 44100, tr, wide)) <= TOL_F64


@pytest.mark.parametrize("hop,n,clips,nmel,ncoef", [(1024, 441000, 3, 128, 20), (1024, 40000, 9, 40, 13), (512, 30001, 2, 128, 40), (777, 20000, 2, 64, 33),
                                                     (1024, 1, 2, 128, 20), (2048, 100000, 2, 13, 12)])
def test_f64_mel_mfcc_on_the_tiled_kernel(zafx, hop, n, clips, nmel, ncoef):
    """W = 2048 in the reference layout, float64, up to 128 filters: k_mel_ft8_f64 (16-frame tiles = 128-byte lines of float64 rows, a frame per
    wavefront, the filterbank over its non-zeros as rows of 64 segments, log + DCT-II rows wave-local) -- whole tiles, an edge tile, rows off
    the line grid (T % 16 != 0), odd hops (the 8-byte load path), more than 32 coefficients, one frame; 1e-12 (mel) / 1e-10 (mfcc) of the
    reference arithmetic.  The other layout and padded rows keep their kernels / pitch."""
    x = np.stack([synth_clip(47, c % 7, n).astype(np.float64) + 1e-9 * (c % 7) for c in range(clips)])
    w = zafx.hamming(2048)
    fb = zafx.melfilterbank(44100, 2048, nmel)
    mel = zafx.melspectrogram_batch(x, w, hop, fb, f64=True)
    assert zafx.mel_plan(w, hop, fb, f64=True).last_kernel == "k_mel_ft8_f64" and mel.dtype == np.float64
    cep = zafx.mfcc_batch(x, w, hop, fb, ncoef, f64=True)
    assert zafx.mel_plan(w, hop, fb, ncoef, f64=True).last_kernel == "k_mel_ft8_f64" and cep.dtype == np.float64
    for c in range(clips):
        ref_mel, ref_cep = orc.melspectrogram(x[c], w, hop, fb), orc.mfcc(x[c], w, hop, fb, ncoef)
        assert mel[c].shape == ref_mel.shape and cep[c].shape == ref_cep.shape
        assert relerr(mel[c], ref_mel) <= TOL_F64, c
        assert relerr(cep[c], ref_cep) <= 1e-10, c
    zafx.melspectrogram_batch(x[:1], w, hop, fb, layout="TF", f64=True)
    assert zafx.mel_plan(w, hop, fb, layout="TF", f64=True).last_kernel == "k_mel_f64"
    # padded rows (row_align) through the plan interface
    pl = zafx.mel_plan(w, hop, fb, ncoef, row_align=16, f64=True)
    got = pl.run_host(x[:2], n)
    assert pl.last_kernel == "k_mel_ft8_f64" and got.shape == cep[:2].shape and np.array_equal(got, cep[:2])


@pytest.mark.parametrize("wl,n", [(2048, 441000), (2048, 30001), (512, 9001), (64, 1000), (8192, 50000), (256, 1)])
def test_f64_mdct_imdct(zafx, wl, n):
    """ZAFX_PRECISION_F64 for the MDCT family: within 1e-12 of the reference arithmetic in both layouts and with padded
    rows; the TDAC round trip reconstructs float64 noise to 1e-12 (the reference's own check, zaf.py:1018-1024)."""
    x = np.stack([synth_clip(37, c, n).astype(np.float64) + 1e-9 * c for c in range(2)])
    w = zafx.kaiser_bessel_derived(wl) if wl != 512 else zafx.sine(wl)
    ref = np.stack([orc.mdct(x[c], w) for c in range(2)])
    for layout in ("FT", "TF"):
        got = zafx.mdct_batch(x, w, layout=layout, f64=True)
        assert got.dtype == np.float64
        if layout == "TF":
            got = got.transpose(0, 2, 1)
        assert got.shape == ref.shape
        for c in range(2):
            assert relerr(got[c], ref[c]) <= TOL_F64, (layout, c)
        coefs = ref if layout == "FT" else np.ascontiguousarray(ref.transpose(0, 2, 1))
        y = zafx.imdct_batch(coefs, w, layout=layout, f64=True)
        assert y.dtype == np.float64
        for c in range(2):
            yref = orc.imdct(ref[c], w)
            assert y[c].shape == yref.shape
            if yref.size:
                assert relerr(y[c], yref) <= TOL_F64, (layout, c)
                m = min(n, yref.size)
                assert np.max(np.abs(y[c][:m] - x[c][:m])) <= 1e-12 * max(1.0, np.max(np.abs(x[c])))   # perfect reconstruction
    p_c, p_p = zafx.mdct_plan(w, f64=True), zafx.mdct_plan(w, f64=True, row_align=8)
    a, b = p_c.run_host(x, n), p_p.run_host(x, n)
    assert a.shape == b.shape and np.array_equal(a, b)


@pytest.mark.parametrize("wl,hop,n", [(2048, 1024, 100000), (1024, 256, 9001), (128, 64, 777), (4096, 1024, 30000)])
def test_magnitude_and_power_spectra(zafx, wl, hop, n):
    """onesided="magnitude" / "power": the spectrogram of the reference's examples,
    np.absolute(audio_stft[0:W/2+1, :]) (zaf.py:83), and its square, as real arrays; both layouts, f32 and f64."""
    x = np.stack([synth_clip(41, c, n) for c in range(2)])
    w = zafx.hamming(wl)
    ref = np.abs(orc.stft_batch(x.astype(np.float64), w, hop)[:, : wl // 2 + 1])
    for f64, tol in ((False, TOL_FFT), (True, 1e-12)):
        for layout in ("FT", "TF"):
            for kind, want in (("magnitude", ref), ("power", ref ** 2)):
                got = zafx.stft_batch(x, w, hop, layout=layout, onesided=kind, f64=f64)
                assert got.dtype == (np.float64 if f64 else np.float32)
                if layout == "TF":
                    got = got.transpose(0, 2, 1)
                assert got.shape == want.shape
                for c in range(2):
                    assert relerr(got[c], want[c]) <= (2 * tol if kind == "power" else tol), (f64, layout, kind, c)
    with pytest.raises(ValueError):
        zafx.istft_batch(np.zeros((1, wl // 2 + 1, 4), np.complex64), w, hop, onesided="magnitude")


@pytest.mark.parametrize("n,clips", [(1024 * 33, 3), (1024 * 49, 2), (1024 * 17, 5)])
def test_spectrogram_kinds_on_rows_of_even_pitch(zafx, n, clips):
    """|X| / |X|^2 at W = 2048 on k_mel2 with a row pitch that is even but not a multiple of 4 (T = 34, 50, 18: the 8-byte
    row pieces of whole tiles, zafx_mel.hip), straight into a device array and into one that starts 8 bytes into an allocation
    (16-byte pieces impossible there whatever T is)."""
    import ctypes
    x = np.stack([synth_clip(59, c, n) for c in range(clips)])
    w = zafx.hamming(2048)
    ref = np.abs(orc.stft_batch(x.astype(np.float64), w, 1024)[:, :1025])
    T = ref.shape[2]
    assert T % 4 == 2 and T >= 18
    d_x = zafx.DeviceBuffer.from_host(x)
    for kind, want in (("magnitude", ref), ("power", ref ** 2)):
        plan = zafx.stft_plan(w, 1024, onesided=kind)
        tol = 2 * TOL_FFT if kind == "power" else TOL_FFT
        d_o = zafx.DeviceBuffer((clips * 1025 * T + 2,), np.float32)
        for shift in (0, 2):      # floats: 0 and 8 bytes
            d_o.fill_zero()
            view = zafx.DeviceBuffer((clips, 1025, T), np.float32, _ptr_from_pool=ctypes.c_void_p(d_o.ptr.value + 4 * shift))
            plan.execute(d_x, view, clips, n)
            plan.sync()
            view.ptr = ctypes.c_void_p()   # (a view, not an allocation: nothing to free)
            assert plan.last_kernel == "k_mel2"
            flat = d_o.download()
            got = flat[shift:shift + clips * 1025 * T].reshape(clips, 1025, T)
            assert not np.any(flat[:shift]) and not np.any(flat[shift + clips * 1025 * T:])   # nothing written outside the array
            for c in range(clips):
                assert relerr(got[c], want[c]) <= tol, (kind, shift, c)
        d_o.free()
    d_x.free()


@pytest.mark.parametrize("dtype,channels,n,clips", [(np.int16, 1, 1024 * 40, 3), (np.int16, 2, 1024 * 40, 3), (np.int16, 1, 30001, 2), (np.int16, 2, 30001, 2),
                                                    (np.int16, 1, 441000, 40), (np.int32, 1, 20000, 2), (np.int16, 5, 20000, 2), (np.int16, 1, 1500, 1)])
def test_execute_pcm_device_resident(zafx, dtype, channels, n, clips):
    """Verdict r4 item 7: integer PCM on the device straight into the transform (Plan.execute_pcm).  mel, mfcc and the |X| / |X|^2 kinds at W = 2048
    read int16 (one or two channels) in k_mel2's own loads, the MDCT at W = 2048 in k_mdct_ft32's (clips of a multiple of four frames), the complex STFT in k_stft_ft16 / k_stft_ft16c's (even clip lengths); the result is BIT-IDENTICAL to normalising first (zaf.py:1202, :65: x / 2^15 and the
    channel mean are exact in float32) -- aligned and odd clip lengths (the sample-by-sample path), more tiles than workgroups.  int32 and other
    channel counts, and every other kind, convert into the plan's staging array first: same numbers as the two-step form."""
    rng = np.random.default_rng([83, channels, n])
    info = np.iinfo(dtype)
    pcm = rng.integers(info.min, info.max, size=(clips, n, channels), endpoint=True).astype(dtype)
    pcm[0, :7] = info.min
    pcm[0, 7:14] = info.max
    w, kbd = zafx.hamming(2048), zafx.kaiser_bessel_derived(2048)
    fb = zafx.melfilterbank(44100, 2048, 128)
    plans = [("mel", zafx.mel_plan(w, 1024, fb)), ("mfcc", zafx.mel_plan(w, 1024, fb, 20)), ("mag", zafx.stft_plan(w, 1024, onesided="magnitude")),
             ("pow", zafx.stft_plan(w, 1024, onesided="power")), ("stft", zafx.stft_plan(w, 1024)), ("stft1", zafx.stft_plan(w, 1024, onesided=True)),
             ("mdct", zafx.mdct_plan(kbd)), ("cqt", zafx.cqt_plan(44100, 25, zafx.cqtkernel(44100, 24, 55, 3520)))]
    if n > 100000:
        plans = plans[:-1]   # (the CQT: short clips only)
    d_pcm = zafx.DeviceBuffer.from_host(pcm)
    d_x = zafx.DeviceBuffer((clips, n), np.float32)
    direct = dtype == np.int16 and channels in (1, 2)
    x64 = (pcm[:1].astype(np.float64) / float(-info.min)).mean(axis=2)[0]
    for name, plan in plans:
        d_a = zafx.DeviceBuffer(plan.out_shape(clips, n), plan.out_dtype)
        d_b = zafx.DeviceBuffer(plan.out_shape(clips, n), plan.out_dtype)
        plan.pcm_to_float(d_pcm, d_x, clips, n, channels)
        plan.execute(d_x, d_a, clips, n)
        plan.sync()
        two_step = d_a.download()
        plan.execute_pcm(d_pcm, d_b, clips, n, channels)
        plan.sync()
        assert plan.last_kernel == {"mel": "k_mel2", "mfcc": "k_mel2", "mag": "k_mel2", "pow": "k_mel2", "mdct": "k_mdct_ft32"}.get(name, plan.last_kernel)
        got = d_b.download()
        assert np.array_equal(got, two_step), (name, direct)
        if name == "mel":
            assert relerr(got[0], orc.melspectrogram(x64, w, 1024, fb)) <= TOL_FB
        d_a.free()
        d_b.free()
    d_pcm.free()
    d_x.free()


@pytest.mark.parametrize("dtype,channels", [(np.int16, 1), (np.int32, 2)])
def test_run_host_pcm_on_a_dct_plan(zafx, dtype, channels):
    """the ZAFX_DCT branch of zafx_run_host_pcm -- integer vectors normalised (zaf.py:1202) and averaged over their
    channels (zaf.py:65) on the device, then zaf.dct type 2 (zaf.py:760-790) on the FFT core."""
    n, rows = 1024, 37
    rng = np.random.default_rng(61)
    info = np.iinfo(dtype)
    pcm = rng.integers(info.min, info.max, size=(rows, n, channels), endpoint=True).astype(dtype)
    x = (pcm.astype(np.float64) / float(-info.min)).mean(axis=2)
    plan = zafx.dct_plan(n, 2)
    got = plan.run_host_pcm(pcm)
    assert plan.last_kernel == "k_dct" and got.shape == (rows, n)
    for r in (0, 1, rows - 1):
        assert relerr(got[r], orc.dct(x[r], 2)) <= TOL_FFT
    assert np.array_equal(got, plan.run_host_pcm(pcm, chunk_clips=5))   # the chunked pipeline: same numbers


def test_concurrent_host_threads(zafx):
    """SURVEY 8b threading contract: the module is re-entrant -- calls from several host threads (ctypes
    releases the GIL; the plan cache is locked; a cached plan serialises its own executions) give the
    same results as serial calls."""
    import threading

    rng_seeds = list(range(6))
    geoms = [(2048, 1024), (1024, 256), (2048, 1024), (512, 128), (2048, 1024), (4096, 2048)]   # some threads share a plan
    xs = [np.stack([synth_clip(50 + s, c, 30000 + 17 * s) for c in range(3)]) for s in rng_seeds]
    want = [orc.stft_batch(x.astype(np.float64), zafx.hamming(wl), hop) for x, (wl, hop) in zip(xs, geoms)]
    got, errors = [None] * len(xs), []

    def work(i):
        try:
            wl, hop = geoms[i]
            for _ in range(3):
                got[i] = zafx.stft_batch(xs[i], zafx.hamming(wl), hop)
                y = zafx.istft_batch(got[i], zafx.hamming(wl), hop)
                assert y.shape[0] == 3
        except Exception as exc:   # surfaced in the main thread
            errors.append((i, repr(exc)))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(xs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i in range(len(xs)):
        for c in range(3):
            assert relerr(got[i][c], want[i][c]) <= TOL_FFT, (i, c)


def test_reference_self_checks(zafx):
    """The four checks the reference's own examples plot instead of asserting: DCT I-IV against
    scipy.fftpack (zaf.py:728-753), the DST inverse pairs (:866-897), imdct(mdct(x)) == x with the Vorbis power-sine
    window (:1098-1109) and istft(stft(x)) == x under COLA (:165-194) -- here on the device."""
    import scipy.fftpack

    rng = np.random.default_rng(99)
    x = rng.standard_normal(1024).astype(np.float32).astype(np.float64)
    for t in (1, 2, 3, 4):
        assert relerr(zafx.dct(x, t), scipy.fftpack.dct(x, type=t, norm="ortho")) <= TOL_FFT
    assert relerr(zafx.dst(zafx.dst(x, 1), 1), x) <= TOL_FFT
    assert relerr(zafx.dst(zafx.dst(x, 2), 3), x) <= TOL_FFT
    assert relerr(zafx.dst(zafx.dst(x, 3), 2), x) <= TOL_FFT
    assert relerr(zafx.dst(zafx.dst(x, 4), 4), x) <= TOL_FFT

    sig = synth_clip(77, 0, 44100).astype(np.float64)
    wl = 2048
    vorbis = np.sin(np.pi / 2 * np.sin(np.pi / wl * (np.arange(wl) + 0.5)) ** 2)   # zaf.py:1101-1104
    rec = zafx.imdct(zafx.mdct(sig, vorbis), vorbis)[: len(sig)]
    assert np.max(np.abs(rec - sig)) < TOL_FFT                                        # BASELINE config 4: residual < 1e-5
    ham = zafx.hamming(wl)
    rec = zafx.istft(zafx.stft(sig, ham, wl // 2), ham, wl // 2)[: len(sig)]
    assert np.max(np.abs(rec - sig)) < TOL_FFT


# ------------------------------------------------------------------ padded rows (row_align)
def _run_padded(zafx, fwd_c, fwd_p, inv_c, inv_p, x, n_in_is_samples=True):
    """Forward then inverse through a compact and a padded plan pair; the padded arrays carry NaN in their padding."""
    b, n = x.shape
    T = fwd_c.out_dims(n)[1]
    pitch = fwd_p.row_pitch(n)
    assert pitch >= T and pitch % fwd_p.row_align == 0 and pitch - T < fwd_p.row_align
    assert fwd_p.out_shape(b, n) == fwd_c.out_shape(b, n)[:2] + (pitch,)
    d_x = zafx.DeviceBuffer.from_host(x)
    d_c = zafx.DeviceBuffer(fwd_c.out_shape(b, n), fwd_c.out_dtype)
    sentinel = np.full(fwd_p.out_shape(b, n), np.nan, dtype=fwd_p.out_dtype)
    d_p = zafx.DeviceBuffer.from_host(sentinel)
    fwd_c.execute(d_x, d_c, b, n)
    fwd_c.sync()
    fwd_p.execute(d_x, d_p, b, n)
    fwd_p.sync()
    compact, padded = d_c.download(), d_p.download()
    # same frames at other addresses (aligned rows take the streaming-store copy of the loop: rounding may differ in the last bit)
    assert compact.size == 0 or relerr(padded[:, :, :T], compact) <= 2e-6
    assert np.isnan(padded[:, :, T:]).all()                    # the padding is never written
    if inv_c is None:
        return
    assert inv_p.row_pitch(T) == pitch
    d_yc = zafx.DeviceBuffer(inv_c.out_shape(b, T), inv_c.out_dtype)
    d_yp = zafx.DeviceBuffer(inv_p.out_shape(b, T), inv_p.out_dtype)
    inv_c.execute(d_c, d_yc, b, T)
    inv_c.sync()
    inv_p.execute(d_p, d_yp, b, T)                             # NaN padding must not reach the output
    inv_p.sync()
    yc, yp = d_yc.download(), d_yp.download()
    # (the padded plan may take the wider-gather variant of the kernel: same frames, other rounding order)
    assert yc.shape == yp.shape and np.isfinite(yp).all() and (yc.size == 0 or relerr(yp, yc) <= 2e-6)


@pytest.mark.parametrize("wl,hop,n,kw", [
    (2048, 1024, 16 * 1024 * 5 + 1024, {}),                    # odd and even frame counts on the persistent kernels
    (2048, 1024, 70001, {}),
    (2048, 512, 50001, {"onesided": True}),
    (1024, 512, 33333, {}),
    (256, 64, 9999, {"onesided": True}),
    (4096, 2048, 70001, {}),                                   # generic one-workgroup-per-tile kernels
    (64, 32, 1000, {}),
    (2048, 1024, 40001, {"f64": True}),
])
def test_row_align_stft_istft(zafx, wl, hop, n, kw):
    ham = zafx.hamming(wl)
    dtype = np.float64 if kw.get("f64") else np.float32
    x = np.stack([synth_clip(7, c, n) for c in range(3)]).astype(dtype)
    for align in (16, 2):
        _run_padded(zafx, zafx.stft_plan(ham, hop, **kw), zafx.stft_plan(ham, hop, row_align=align, **kw),
                    zafx.istft_plan(ham, hop, **kw), zafx.istft_plan(ham, hop, row_align=align, **kw), x)


@pytest.mark.parametrize("wl,n", [(2048, 70001), (2048, 1024 * 33), (512, 20001), (8192, 70001), (64, 1000)])
def test_row_align_mdct_imdct(zafx, wl, n):
    kbd = zafx.kaiser_bessel_derived(wl)
    x = np.stack([synth_clip(7, c, n) for c in range(3)]).astype(np.float32)
    for align in (32, 4):
        _run_padded(zafx, zafx.mdct_plan(kbd), zafx.mdct_plan(kbd, row_align=align),
                    zafx.mdct_plan(kbd, inverse=True), zafx.mdct_plan(kbd, inverse=True, row_align=align), x)


def test_row_align_mel_cqt_and_host_path(zafx):
    ham = zafx.hamming(2048)
    fb = zafx.melfilterbank(44100, 2048, 128)
    x = np.stack([synth_clip(7, c, 50001) for c in range(2)]).astype(np.float32)
    _run_padded(zafx, zafx.mel_plan(ham, 1024, fb), zafx.mel_plan(ham, 1024, fb, row_align=32), None, None, x)
    _run_padded(zafx, zafx.mel_plan(ham, 1024, fb, 20), zafx.mel_plan(ham, 1024, fb, 20, row_align=32), None, None, x)
    ck = zafx.cqtkernel(44100, 24, 55, 3520)
    _run_padded(zafx, zafx.cqt_plan(44100, 25, ck), zafx.cqt_plan(44100, 25, ck, row_align=32), None, None, x)
    _run_padded(zafx, zafx.cqt_plan(44100, 25, ck, 24), zafx.cqt_plan(44100, 25, ck, 24, row_align=32), None, None, x)
    # host-array path: compact arrays in and out whatever the device pitch is
    p_c, p_p = zafx.stft_plan(ham, 1024), zafx.stft_plan(ham, 1024, row_align=16)
    s_c, s_p = p_c.run_host(x, x.shape[1]), p_p.run_host(x, x.shape[1])
    assert s_p.shape == s_c.shape and relerr(s_p, s_c) <= 2e-6
    i_c, i_p = zafx.istft_plan(ham, 1024), zafx.istft_plan(ham, 1024, row_align=16)
    assert relerr(i_p.run_host(s_c, s_c.shape[2]), i_c.run_host(s_c, s_c.shape[2])) <= 2e-6
    with pytest.raises(ValueError):
        zafx.stft_plan(ham, 1024, layout="TF", row_align=16)
    with pytest.raises(ValueError):
        zafx.stft_plan(ham, 1024, row_align=24)


# ------------------------------------------------------------------ few long clips: the carry kernels cut a clip into segments
@pytest.mark.parametrize("n", [44100 * 30, 44100 * 30 + 12345])
def test_long_clips_are_segmented(zafx, n):
    """3 clips x 30 s: fewer clips than CUs, so carry_segments() splits every clip into many segments, each started by a
    carry-only tile -- inverse transforms of both layouts against the oracle, and the round trip."""
    x = np.stack([synth_clip(47, c, n) for c in range(3)])
    ham, kbd = zafx.hamming(2048), zafx.kaiser_bessel_derived(2048)
    spec = zafx.stft_batch(x, ham, 1024)
    coef = zafx.mdct_batch(x, kbd)
    y_ref = [orc.istft(spec[c].astype(np.complex128), ham, 1024) for c in range(3)]
    z_ref = [orc.imdct(coef[c].astype(np.float64), kbd) for c in range(3)]
    for layout in ("FT", "TF"):
        s_in = spec if layout == "FT" else np.ascontiguousarray(spec.transpose(0, 2, 1))
        c_in = coef if layout == "FT" else np.ascontiguousarray(coef.transpose(0, 2, 1))
        y = zafx.istft_batch(s_in, ham, 1024, layout=layout)
        z = zafx.imdct_batch(c_in, kbd, layout=layout)
        half = spec[:, :1025] if layout == "FT" else np.ascontiguousarray(spec[:, :1025].transpose(0, 2, 1))
        y1 = zafx.istft_batch(np.ascontiguousarray(half), ham, 1024, layout=layout, onesided=True)
        for c in range(3):
            assert y[c].shape == y_ref[c].shape and relerr(y[c], y_ref[c]) <= TOL_FFT, (layout, c)
            assert relerr(y1[c], y_ref[c]) <= TOL_FFT, (layout, c)
            assert z[c].shape == z_ref[c].shape and relerr(z[c], z_ref[c]) <= TOL_FFT, (layout, c)
            assert np.max(np.abs(y[c][:n] - x[c])) < 1e-4 and np.max(np.abs(z[c][:n] - x[c])) < 1e-4


def test_clip_longer_than_32_bit_byte_offsets(zafx):
    """One clip of 2^29 + 4098 samples (3.4 hours at 44.1 kHz, 2 GB): the aligned forms of k_mel / k_mdct_ft32 address a clip through
    a buffer descriptor with 32-bit byte offsets and hand such a clip to the forms with 64-bit addresses (from 2^29 samples on; the
    descriptor itself ends at 4 GB).  Checked on the last
    frames (the oracle runs on the tail of the signal: frame j0 + i of the clip is frame i of x[j0 * hop:], but for the padded i = 0)."""
    n, hop, wl = (1 << 29) + 4098, 1024, 2048
    block = synth_clip(71, 0, 1000003)
    x = np.resize(block, n)[None, :]
    ham = zafx.hamming(wl)
    fb = zafx.melfilterbank(44100, wl, 128)
    t = -(-n // hop) + 1
    j0 = t - 40
    tail = x[0, j0 * hop:].astype(np.float64)
    mel = zafx.melspectrogram_batch(x, ham, hop, fb)
    assert mel.shape == (1, 128, t)
    assert relerr(mel[0][:, j0 + 1:], orc.melspectrogram(tail, ham, hop, fb)[:, 1:]) <= TOL_FB
    del mel
    cep = zafx.mfcc_batch(x, ham, hop, fb, 20)
    assert relerr(cep[0][:, j0 + 1:], orc.mfcc(tail, ham, hop, fb, 20)[:, 1:]) <= TOL_FB
    del cep
    kbd = zafx.kaiser_bessel_derived(wl)
    coef = zafx.mdct_batch(x, kbd)   # frames of hop 1024 starting one hop ahead of the clip (zaf.py:1036-1041)
    tm = coef.shape[2]
    ref = orc.mdct(x[0, (tm - 40) * hop:].astype(np.float64), kbd)
    assert relerr(coef[0][:, tm - 40 + 1:tm - 40 + ref.shape[1]], ref[:, 1:]) <= TOL_FFT


@pytest.mark.parametrize("nmel,ncoef,hop,n,clips,seed", [
    (128, 20, 1024, 441000, 3, 0), (40, 13, 512, 30001, 5, 1), (1, None, 1024, 2048, 2, 2), (17, 16, 1000, 50000, 1, 3),
    (128, 32, 2048, 100000, 2, 4), (100, 33, 1024, 40000, 2, 5), (200, 20, 1024, 40000, 2, 6), (64, 5, 777, 12345, 260, 7)])
def test_mel2_geometries(zafx, nmel, ncoef, hop, n, clips, seed):
    """k_mel2 (W = 2048: the filterbank product of a tile under the next tile's transforms; whole-block items, cut blocks, the mfcc stage
    between its barriers) over filter counts that are not multiples of 16, one filter, more coefficient rows than one block, more than 128
    filters / 32 coefficients (mfcc then runs k_mel), odd hops and clip lengths (the unaligned form), fewer tiles than workgroups, more clips
    than workgroups, both layouts."""
    x = np.stack([synth_clip(90 + seed, c % 7, n) for c in range(clips)])
    w = zafx.hamming(2048)
    fb = zafx.melfilterbank(44100, 2048, nmel)
    probe = sorted({0, clips // 2, clips - 1})
    mel = zafx.melspectrogram_batch(x, w, hop, fb)
    for c in probe:
        ref = orc.melspectrogram(x[c].astype(np.float64), w, hop, fb)
        assert mel[c].shape == ref.shape and relerr(mel[c], ref) <= TOL_FB, ("mel", c)
    mel_tf = zafx.melspectrogram_batch(x[:2], w, hop, fb, layout="TF")
    assert np.array_equal(mel_tf.transpose(0, 2, 1), mel[:2])
    if ncoef is not None:
        cep = zafx.mfcc_batch(x, w, hop, fb, ncoef)
        for c in probe:
            ref = orc.mfcc(x[c].astype(np.float64), w, hop, fb, ncoef)
            assert cep[c].shape == ref.shape and relerr(cep[c], ref) <= TOL_FB, ("mfcc", c)
        cep_tf = zafx.mfcc_batch(x[:2], w, hop, fb, ncoef, layout="TF")
        assert np.array_equal(cep_tf.transpose(0, 2, 1), cep[:2])


@pytest.mark.parametrize("wl,hop,n", [(2048, 3000, 50000), (2048, 2049, 20001), (1024, 5000, 30000), (256, 1000, 9999), (4096, 6000, 40000)])
def test_hop_above_window(zafx, wl, hop, n):
    """step_length > window_length: zaf.stft / melspectrogram / mfcc skip samples between frames (zaf.py:102-136 holds for
    any hop); the forward kernels follow, the ISTFT (whose reference trims by a negative amount then) refuses."""
    x = np.stack([synth_clip(53, c, n) for c in range(2)])
    w = zafx.hamming(wl)
    for layout in ("FT", "TF"):
        for one in (False, True):
            got = zafx.stft_batch(x, w, hop, layout=layout, onesided=one)
            if layout == "TF":
                got = got.transpose(0, 2, 1)
            for c in range(2):
                ref = orc.stft(x[c].astype(np.float64), w, hop)
                ref = ref[:wl // 2 + 1] if one else ref
                assert got[c].shape == ref.shape and relerr(got[c], ref) <= TOL_FFT, (layout, one, c)
    if wl <= 2048:
        fb = zafx.melfilterbank(44100, wl, 40)
        mel = zafx.melspectrogram_batch(x, w, hop, fb)
        for c in range(2):
            assert relerr(mel[c], orc.melspectrogram(x[c].astype(np.float64), w, hop, fb)) <= TOL_FB
    got64 = zafx.stft_batch(x.astype(np.float64), w, hop, f64=True)
    assert relerr(got64[0], orc.stft(x[0].astype(np.float64), w, hop)) <= 1e-12
    with pytest.raises((ValueError, zafx.ZafxError)):
        zafx.istft_plan(w, hop)


def test_cqt_long_kernel(zafx):
    """A lower minimum frequency makes fft_length 65536 (zaf.py:505-509): more than LDS holds as float32 pairs, so the
    transform runs on the float64 kernel (which decimates the frame) whatever precision was asked for."""
    fs, res, tr, n = 44100, 24, 20, 80000
    ck = zafx.cqtkernel(fs, res, 27.5, 1760)
    assert ck.shape[1] == 65536
    x = np.stack([synth_clip(59, c, n) for c in range(2)])
    got = zafx.cqtspectrogram_batch(x, fs, tr, ck)
    got64 = zafx.cqtspectrogram_batch(x.astype(np.float64), fs, tr, ck, f64=True)
    chroma = zafx.cqtchromagram_batch(x, fs, tr, res, ck)
    assert got.dtype == np.float32 and got64.dtype == np.float64
    for c in range(2):
        ref = orc.cqtspectrogram(x[c].astype(np.float64), fs, tr, ck)
        assert got[c].shape == ref.shape
        assert relerr(got64[c], ref) <= 1e-12 and relerr(got[c], ref) <= TOL_FB
        assert relerr(chroma[c], orc.cqtchromagram(x[c].astype(np.float64), fs, tr, res, ck)) <= TOL_FB
    one = zafx.cqtspectrogram(x[0], fs, tr, ck)
    assert one.dtype == np.float64 and relerr(one, orc.cqtspectrogram(x[0].astype(np.float64), fs, tr, ck)) <= TOL_FB


@pytest.mark.parametrize("fs,res,fmin,fmax,tr", [(16000, 2, 220.0, 7040.0, 100),   # fft_length 256
                                                (8000, 3, 440.0, 3520.0, 100),     # 128
                                                (16000, 2, 220.0, 440.0, 200)])    # 256, two bins
def test_cqt_short_kernel(zafx, fs, res, fmin, fmax, tr):
    """Few bins per octave over a high minimum frequency give an fft_length below the 512 samples of the shortest device frame
    (zaf.py:505-509); the host rewrites the kernel for 512-sample frames (core._cqt_embed) -- same numbers as the reference."""
    ck = zafx.cqtkernel(fs, res, fmin, fmax)
    assert ck.shape[1] < 512
    x = np.stack([synth_clip(67, c, 20001) for c in range(3)])
    got = zafx.cqtspectrogram_batch(x, fs, tr, ck)
    got_tf = zafx.cqtspectrogram_batch(x, fs, tr, ck, layout="TF")
    got64 = zafx.cqtspectrogram_batch(x.astype(np.float64), fs, tr, ck, f64=True)
    chroma = zafx.cqtchromagram_batch(x, fs, tr, res, ck)
    for c in range(3):
        ref = orc.cqtspectrogram(x[c].astype(np.float64), fs, tr, ck)
        assert got[c].shape == ref.shape
        assert relerr(got[c], ref) <= TOL_FB and relerr(got_tf[c].T, ref) <= TOL_FB and relerr(got64[c], ref) <= 1e-12
        assert relerr(chroma[c], orc.cqtchromagram(x[c].astype(np.float64), fs, tr, res, ck)) <= TOL_FB
    with pytest.raises(ValueError):   # step above fft_length: the reference's padding is negative there (np.pad raises)
        zafx.cqtspectrogram_batch(x, fs, fs // (ck.shape[1] + 64), ck)


@pytest.mark.parametrize("wl,hop,nmel", [(4096, 2048, 128), (8192, 2048, 64), (2048, 1024, 300), (4096, 1024, 40), (8192, 4096, 256), (4096, 1763, 128), (4096, 2048, 400),
                                         (1024, 512, 500)])
def test_mel_long_window_or_wide_bank(zafx, wl, hop, nmel):
    """Windows of 4096 / 8192 samples are outside the fused W <= 2048 kernel.  W = 4096: k_mel_ft16b, the two-band STFT kernel with the
    filterbank product in place of the stores; W = 8192: a spectrum kernel (|X| or |X|^2 rows into a plan-owned scratch) + the
    banded filterbank kernel k_melfb; both float32.  Filterbanks of 257 ... 576 rows take the k_melfb route at any window."""
    x = np.stack([synth_clip(61, c, 60000 + (hop % 2)) for c in range(2)])
    w = zafx.hamming(wl)
    fb = zafx.melfilterbank(44100, wl, nmel)
    plan = zafx.mel_plan(w, hop, fb)
    assert plan.kernel_name == ("k_melfb" if nmel > 256 else {4096: "k_mel_ft16b", 8192: "k_melfb"}[wl]) and plan.in_dtype == np.float32
    mel = zafx.melspectrogram_batch(x, w, hop, fb)
    cep = zafx.mfcc_batch(x, w, hop, fb, 13)
    assert mel.dtype == np.float32 and cep.dtype == np.float32
    for c in range(2):
        x64 = x[c].astype(np.float64)
        assert relerr(mel[c], orc.melspectrogram(x64, w, hop, fb)) <= TOL_FB
        assert relerr(cep[c], orc.mfcc(x64, w, hop, fb, 13)) <= TOL_FB
    mel_tf = zafx.melspectrogram_batch(x, w, hop, fb, layout="TF")
    cep_tf = zafx.mfcc_batch(x, w, hop, fb, 13, layout="TF")
    assert np.array_equal(mel_tf.transpose(0, 2, 1), mel) and np.array_equal(cep_tf.transpose(0, 2, 1), cep)
    one = zafx.melspectrogram(x[0], w, hop, fb)
    assert one.dtype == np.float64 and relerr(one, orc.melspectrogram(x[0].astype(np.float64), w, hop, fb)) <= TOL_FB


@pytest.mark.parametrize("fs,wl,hop,nmel,n", [(16000, 400, 160, 80, 48000), (44100, 1764, 441, 128, 60001), (44100, 3000, 1500, 64, 70000),
                                              (16000, 402, 160, 40, 20000), (8000, 200, 80, 23, 9999), (44100, 6000, 2000, 256, 50000)])
def test_mel_window_not_a_power_of_two(zafx, fs, wl, hop, nmel, n):
    """melspectrogram / mfcc with windows that are not a power of two (the 25 ms / 10 ms framing of speech front ends: 400 / 160
    samples at 16 kHz) run in float32: the Bluestein STFT's magnitude / power kind into the plan's scratch + k_melfb; they ran on the
    float64 kernel.  zaf.py:369-373, :436-452."""
    x = np.stack([synth_clip(71, c, n) for c in range(3)])
    w = zafx.hamming(wl)
    fb = zafx.melfilterbank(fs, wl, nmel)
    plan = zafx.mel_plan(w, hop, fb)
    assert plan.kernel_name == "k_melfb" and plan.in_dtype == np.float32
    mel = zafx.melspectrogram_batch(x, w, hop, fb)
    cep = zafx.mfcc_batch(x, w, hop, fb, 13)
    mel_tf = zafx.melspectrogram_batch(x, w, hop, fb, layout="TF")
    assert mel.dtype == np.float32 and np.array_equal(mel_tf.transpose(0, 2, 1), mel)
    for c in range(3):
        x64 = x[c].astype(np.float64)
        assert relerr(mel[c], orc.melspectrogram(x64, w, hop, fb)) <= TOL_FB, c
        assert relerr(cep[c], orc.mfcc(x64, w, hop, fb, 13)) <= TOL_FB, c
    one = zafx.mfcc(x[0], w, hop, fb, 13)
    assert one.dtype == np.float64 and relerr(one, orc.mfcc(x[0].astype(np.float64), w, hop, fb, 13)) <= TOL_FB


@pytest.mark.parametrize("wl", [8192, 4096])
def test_mel_long_window_in_chunks(zafx, wl):
    """k_melfb's scratch holds a chunk of clips (256 MB): 1200 short clips at W = 8192 go through it in two chunks;
    padded output rows (row_align).  (W = 4096 runs the fused k_mel_ft16b: same checks, many tiles per workgroup.)"""
    n, hop, clips = 30000, wl // 2, 1200
    x = np.stack([synth_clip(62, c % 7, n) for c in range(clips)])
    w = zafx.hamming(wl)
    fb = zafx.melfilterbank(44100, wl, 128)
    mel = zafx.melspectrogram_batch(x, w, hop, fb)
    cep = zafx.mfcc_batch(x, w, hop, fb, 20)
    for c in (0, 255, 767, 768, 1023, 1199):
        x64 = x[c].astype(np.float64)
        assert relerr(mel[c], orc.melspectrogram(x64, w, hop, fb)) <= TOL_FB, c
        assert relerr(cep[c], orc.mfcc(x64, w, hop, fb, 20)) <= TOL_FB, c
    assert np.array_equal(mel[7:14], mel[0:7]) and np.array_equal(cep[1190:1197], cep[0:7])
    _run_padded(zafx, zafx.mel_plan(w, hop, fb), zafx.mel_plan(w, hop, fb, row_align=32), None, None, x[:3])
    _run_padded(zafx, zafx.mel_plan(w, hop, fb, 20), zafx.mel_plan(w, hop, fb, 20, row_align=32), None, None, x[:3])


@pytest.mark.parametrize("wl,hop,n", [(2048, 100, 20000), (4096, 300, 30000), (8192, 1000, 40000), (64, 3, 1000)])
def test_istft_tiny_hop(zafx, wl, hop, n):
    """More overlapping frames per sample than the float32 overlap-add tile holds: up to W = 2048 the plan takes the float32
    frames + gather overlap-add form (zafx_bs32.hip), above it the host layer runs the ISTFT on the float64 kernels (no such
    limit either) and still returns float32."""
    x = synth_clip(67, 0, n)
    w = zafx.hamming(wl)
    ref_s = orc.stft(x.astype(np.float64), w, hop)
    y = zafx.istft_batch(ref_s[None].astype(np.complex64), w, hop)[0]
    yref = orc.istft(ref_s, w, hop)
    assert y.dtype == np.float32 and y.shape == yref.shape and relerr(y, yref) <= TOL_FFT
    y1 = zafx.istft(ref_s, w, hop)
    assert y1.dtype == np.float64 and relerr(y1, yref) <= TOL_FFT


@pytest.mark.parametrize("wl,hop,n", [(1000, 500, 20000), (1764, 441, 30000), (777, 300, 9999), (2047, 1024, 40000), (6, 3, 100), (3, 1, 50),
                                      (32, 8, 500), (16, 16, 300)])
def test_window_not_a_power_of_two(zafx, wl, hop, n):
    """np.fft takes any length, so zaf.stft / istft / melspectrogram / mfcc take any window: lengths that are not a power
    of two (up to 2048) run on the float64 Bluestein kernels, selected by the host layer (even and odd lengths)."""
    x = np.stack([synth_clip(71, c, n) for c in range(2)])
    w = orc.hamming_periodic(wl) if wl > 3 else np.array([0.5, 1.0, 0.5])
    for layout in ("FT", "TF"):
        for one in (False, True):
            got = zafx.stft_batch(x, w, hop, layout=layout, onesided=one)
            assert got.dtype == np.complex64
            if layout == "TF":
                got = got.transpose(0, 2, 1)
            for c in range(2):
                ref = orc.stft(x[c].astype(np.float64), w, hop)
                want = ref[:wl // 2 + 1] if one else ref
                assert got[c].shape == want.shape and relerr(got[c], want) <= TOL_FFT, (layout, one, c)
                s_in = want if layout == "FT" else np.ascontiguousarray(want.T)
                y = zafx.istft_batch(s_in[None], w, hop, layout=layout, onesided=one, f64=True)[0]
                yref = orc.istft(ref, w, hop)
                assert y.shape == yref.shape and (yref.size == 0 or relerr(y, yref) <= 1e-11), (layout, one, c)
                y32 = zafx.istft_batch(s_in[None], w, hop, layout=layout, onesided=one)[0]   # float32 Bluestein form for wl >= 33
                assert y32.dtype == np.float32 and y32.shape == yref.shape and (yref.size == 0 or relerr(y32, yref) <= TOL_FFT), (layout, one, c)
    got64 = zafx.stft_batch(x.astype(np.float64), w, hop, f64=True)
    assert got64.dtype == np.complex128 and relerr(got64[0], orc.stft(x[0].astype(np.float64), w, hop)) <= 1e-11
    mag = zafx.stft_batch(x, w, hop, onesided="magnitude")
    assert mag.dtype == np.float32 and relerr(mag[0], np.abs(orc.stft(x[0].astype(np.float64), w, hop)[:wl // 2 + 1])) <= TOL_FFT
    try:
        fb = zafx.melfilterbank(44100, wl, 20) if wl >= 64 else None
    except ValueError:   # (zaf.melfilterbank itself fails for some odd lengths, e.g. 2047: same exception here)
        fb = None
    if fb is not None:
        mel = zafx.melspectrogram_batch(x, w, hop, fb)
        cep = zafx.mfcc_batch(x, w, hop, fb, 8)
        for c in range(2):
            x64 = x[c].astype(np.float64)
            assert relerr(mel[c], orc.melspectrogram(x64, w, hop, fb)) <= TOL_FB
            assert relerr(cep[c], orc.mfcc(x64, w, hop, fb, 8)) <= TOL_FB
    one_clip = zafx.stft(x[0], w, hop)
    assert one_clip.dtype == np.complex128 and relerr(one_clip, orc.stft(x[0].astype(np.float64), w, hop)) <= TOL_FFT


@pytest.mark.parametrize("wl,n", [(1920, 30000), (1152, 20001), (1000, 9999), (6, 100), (2046, 40000), (32, 500), (8, 100)])
def test_mdct_window_not_a_power_of_two(zafx, wl, n):
    """Even window lengths that are not a power of two (AAC's 1920, MP3's 1152 ...): the reference's own W-point FFT
    formulation through the float64 Bluestein kernels; TDAC round trip with a sine window."""
    x = np.stack([synth_clip(73, c, n) for c in range(2)])
    w = orc.sine_window(wl)
    for layout in ("FT", "TF"):
        got = zafx.mdct_batch(x, w, layout=layout)
        got64 = zafx.mdct_batch(x.astype(np.float64), w, layout=layout, f64=True)
        assert got.dtype == np.float32 and got64.dtype == np.float64
        if layout == "TF":
            got, got64 = got.transpose(0, 2, 1), got64.transpose(0, 2, 1)
        for c in range(2):
            ref = orc.mdct(x[c].astype(np.float64), w)
            assert got[c].shape == ref.shape and relerr(got[c], ref) <= TOL_FFT and relerr(got64[c], ref) <= 1e-11, (layout, c)
            c_in = ref if layout == "FT" else np.ascontiguousarray(ref.T)
            y = zafx.imdct_batch(c_in[None], w, layout=layout, f64=True)[0]
            yref = orc.imdct(ref, w)
            assert y.shape == yref.shape and (yref.size == 0 or relerr(y, yref) <= 1e-11), (layout, c)
            y32 = zafx.imdct_batch(c_in[None], w, layout=layout)[0]   # float32 Bluestein form for wl >= 34
            assert y32.dtype == np.float32 and y32.shape == yref.shape and (yref.size == 0 or relerr(y32, yref) <= TOL_FFT), (layout, c)
            m = min(n, yref.size)
            assert m == 0 or np.max(np.abs(y[:m] - x[c][:m])) < 1e-9
    with pytest.raises(ValueError):
        zafx.mdct_batch(x, np.ones(999))


def test_bluestein_f32_plans_and_sizes(zafx):
    """The float32 Bluestein forms (zafx_bs32.hip): which plans take them, every convolution length 128 ... 4096, a hop far
    below the window (the gather overlap-add has no tile limit), a batch, kernel names."""
    for wl, hop, name in ((33, 11, "k_stft_bs32"), (100, 25, "k_stft_bs32"), (1764, 441, "k_stft_bs32"), (2047, 100, "k_stft_bs32")):
        w = orc.hamming_periodic(wl)
        plan = zafx.stft_plan(w, hop)
        assert not plan.f64 and plan.kernel_name == name and plan.out_dtype == np.complex64
        assert zafx.istft_plan(w, hop).kernel_name == "k_ifft_frames_bs32"
    assert zafx.stft_plan(orc.hamming_periodic(31), 7).f64            # below 33 samples: float64 kernels
    assert zafx.mdct_plan(orc.sine_window(1920)).kernel_name == "k_mdct_bs32"
    assert zafx.mdct_plan(orc.sine_window(1920), inverse=True).kernel_name == "k_imdct_frames_bs32"
    x = np.stack([synth_clip(75, c, 30000) for c in range(5)])
    for wl, hop in ((65, 16), (129, 64), (300, 75), (600, 200), (1100, 275), (1764, 441), (2047, 100)):   # M = 256 ... 4096
        w = orc.hamming_periodic(wl)
        got = zafx.stft_batch(x, w, hop)
        ref = orc.stft(x[4].astype(np.float64), w, hop)
        assert got[4].shape == ref.shape and relerr(got[4], ref) <= TOL_FFT, (wl, hop)
        y = zafx.istft_batch(got, w, hop)
        yref = orc.istft(ref, w, hop)
        assert relerr(y[4], yref) <= 3 * TOL_FFT, (wl, hop)   # (round trip: the float32 spectrum goes back in)


@pytest.mark.gpu
@pytest.mark.parametrize("wl,hop", [(3000, 750), (4410, 2205), (5000, 1250), (8000, 2000), (6001, 1500)])
def test_bluestein_f32_above_2048(zafx, wl, hop):
    """Windows of 2049 ... 8192 samples that are not a power of two (np.fft takes any length, zaf.py:139): float32 Bluestein
    forms with convolution lengths 8192 and 16384, STFT / ISTFT for any length, MDCT / IMDCT for the even ones; and the
    powers of two 4096 / 8192 with a hop too small for the tiled overlap-add, which take the same frames + gather form."""
    x = np.stack([synth_clip(76, c, 40000) for c in range(3)])
    w = orc.hamming_periodic(wl)
    plan = zafx.stft_plan(w, hop)
    assert not plan.f64 and plan.kernel_name == "k_stft_bs32"
    got = zafx.stft_batch(x, w, hop)
    ref = orc.stft(x[2].astype(np.float64), w, hop)
    assert got[2].shape == ref.shape and relerr(got[2], ref) <= TOL_FFT
    y = zafx.istft_batch(got, w, hop)
    assert relerr(y[2], orc.istft(ref, w, hop)) <= 3 * TOL_FFT
    if wl % 2 == 0:
        ws = orc.sine_window(wl)
        assert zafx.mdct_plan(ws).kernel_name == "k_mdct_bs32"
        c = zafx.mdct_batch(x, ws)
        cref = orc.mdct(x[2].astype(np.float64), ws)
        assert c[2].shape == cref.shape and relerr(c[2], cref) <= TOL_FFT
        yi = zafx.imdct_batch(c, ws)
        assert relerr(yi[2], orc.imdct(cref, ws)) <= 3 * TOL_FFT


@pytest.mark.gpu
@pytest.mark.parametrize("wl,hop", [(4096, 64), (8192, 1000)])
def test_istft_small_hop_large_window_f32(zafx, wl, hop):
    w = orc.hamming_periodic(wl)
    plan = zafx.istft_plan(w, hop)
    assert not plan.f64 and plan.kernel_name == "k_ifft_frames_bs32"
    x = synth_clip(77, 0, 30000)
    spec = orc.stft(x.astype(np.float64), w, hop)
    y = zafx.istft(spec.astype(np.complex64), w, hop)
    assert relerr(y, orc.istft(spec, w, hop)) <= 3 * TOL_FFT


def test_f64_inverse_transforms_in_scratch_chunks(tmp_path):
    """The float64 ISTFT / IMDCT park their time-domain frames in a plan-owned scratch; batches above the budget (1 GiB,
    here forced to 1 MiB) go through it in chunks of clips and must give the same samples."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    code = f"""
import sys, numpy as np
sys.path.insert(0, {os.path.join(ROOT, 'zaf-python_amd')!r}); sys.path.insert(0, {ROOT!r})
import zafx
from oracle import zaf_oracle as orc
x = np.stack([np.random.default_rng([31, c]).standard_normal(30001) for c in range(7)])
w = zafx.hamming(1024)
for layout in ("FT", "TF"):
    S = zafx.stft_batch(x, w, 256, layout=layout, f64=True)
    y = zafx.istft_batch(S, w, 256, layout=layout, f64=True)          # 7 clips x 122 frames x 1024 x 8 B = 7 MB of scratch: 7 chunks
    for c in range(7):
        ref = orc.istft(orc.stft(x[c], w, 256), w, 256)
        assert np.max(np.abs(y[c] - ref)) <= 1e-11 * np.max(np.abs(ref)), (layout, c)
kbd = zafx.kaiser_bessel_derived(1024)
m = zafx.mdct_batch(x, kbd, f64=True)
r = zafx.imdct_batch(m, kbd, f64=True)
for c in range(7):
    ref = orc.imdct(orc.mdct(x[c], kbd), kbd)
    assert np.max(np.abs(r[c] - ref)) <= 1e-11 * np.max(np.abs(ref)), c
w2 = zafx.hamming(1000)                                                  # Bluestein forms
y2 = zafx.istft_batch(zafx.stft_batch(x, w2, 250), w2, 250)
assert np.max(np.abs(y2[3] - orc.istft(orc.stft(x[3], w2, 250), w2, 250))) < 1e-5
print("chunked ok")
"""
    env = dict(os.environ, ZAFX_SCRATCH_BUDGET_MB="1")
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, timeout=300)
    assert res.returncode == 0 and b"chunked ok" in res.stdout, res.stderr.decode()[-2000:]


def test_host_layer_fixes_of_round_2(zafx):
    """run_host casts any real dtype to the plan's own (it used to reinterpret the bytes); the allocation pool is
    bounded and evicts its oldest entries; the dct / dst plans are keyed before their N x N matrix is built."""
    from zafx import core
    x = np.stack([synth_clip(2, c, 5000) for c in range(2)])
    ham = zafx.hamming(256)
    plan = zafx.stft_plan(ham, 64)
    ref = plan.run_host(x, x.shape[1])
    assert np.array_equal(plan.run_host(x.astype(np.float64), x.shape[1]), ref)
    assert np.array_equal(plan.run_host((x * 1000).astype(np.int32), x.shape[1]), plan.run_host(np.trunc(x * 1000).astype(np.float32), x.shape[1]))
    with pytest.raises(ValueError):
        plan.run_host(x.astype(np.complex64), x.shape[1])
    # pool: never above its cap, oldest size class evicted first
    core.DeviceBuffer.drain_pool()
    cap = core.DeviceBuffer._POOL_CAP
    try:
        core.DeviceBuffer._POOL_CAP = 3 << 20
        for mb in (1, 1, 1, 2, 1):
            core.DeviceBuffer((mb << 20,), np.uint8).release()
        assert core.DeviceBuffer._pool_bytes[0] <= 3 << 20
        assert sum(len(v) * k[1] for k, v in core.DeviceBuffer._pool.items()) == core.DeviceBuffer._pool_bytes[0]
        big = core.DeviceBuffer((4 << 20,), np.uint8)
        big.release()                                   # larger than the whole pool: freed, not parked
        assert core.DeviceBuffer._pool_bytes[0] <= 3 << 20
    finally:
        core.DeviceBuffer._POOL_CAP = cap
        core.DeviceBuffer.drain_pool()
    # dct: the second call finds the plan by (transform, type, N, device) without touching the matrix builder
    v = synth_clip(3, 0, 300)
    first = zafx.dct(v, 2)
    calls = []
    orig = zafx.constants.dct_matrix if hasattr(zafx, "constants") else None
    import zafx.constants as zc
    orig = zc.dct_matrix
    try:
        def spy(n, t):
            calls.append((n, t))
            return orig(n, t)
        spy.__name__ = "dct_matrix"
        zc.dct_matrix = spy
        assert np.array_equal(zafx.dct(v, 2), first) and not calls
    finally:
        zc.dct_matrix = orig
    with pytest.raises(ValueError):
        zafx.dct(v, 5)


@pytest.mark.parametrize("n_clips,n", [(1, 40000), (3, 52345), (9, 36000), (17, 33000)])
def test_cqt_clip_groups_and_complex_kernels(zafx, n_clips, n):
    """k_cqt deals the frames of a group of clips (an XCD's share) round-robin to its workgroups: batches that are not a multiple
    of 8 clips, fewer clips than XCDs, fewer frames than workgroups.  Kernel matrices beyond the reference's own: complex values
    (per-row phase: the streamed complex form), scattered columns on both halves of the spectrum (conjugate entries), an empty row,
    both layouts, chromagram."""
    x = np.stack([synth_clip(21, c, n) for c in range(n_clips)])
    ck = zafx.cqtkernel(44100, 24, 55, 3520)                       # real, resident form
    got = zafx.cqtspectrogram_batch(x, 44100, 25, ck)
    for c in sorted({0, n_clips // 2, n_clips - 1}):
        assert relerr(got[c], orc.cqtspectrogram(x[c].astype(np.float64), 44100, 25, ck)) <= TOL_FB, c
    got_tf = zafx.cqtspectrogram_batch(x, 44100, 25, ck, layout="TF")
    assert np.array_equal(got_tf.transpose(0, 2, 1), got)
    ch = zafx.cqtchromagram_batch(x[:2], 44100, 25, 24, ck)
    assert relerr(ch[0], orc.cqtchromagram(x[0].astype(np.float64), 44100, 25, 24, ck)) <= TOL_FB
    # complex values: every row times a unit phase (the magnitudes stay, the arithmetic is complex x complex)
    rng = np.random.default_rng(5)
    csr_k = ck.tocsr()
    phase = np.exp(2j * np.pi * rng.random(csr_k.shape[0]))
    kc = scipy.sparse.csr_matrix(scipy.sparse.diags(phase) @ csr_k)
    gc = zafx.cqtspectrogram_batch(x[:2], 44100, 25, kc)
    assert relerr(gc[-1], orc.cqtspectrogram(x[:2][-1].astype(np.float64), 44100, 25, kc)) <= TOL_FB
    # scattered columns on both halves, complex values, one empty row, rows longer than a wave
    rows, cols = 37, 4096
    dense = np.zeros((rows, cols), dtype=np.complex128)
    for r in range(rows):
        if r == 5:
            continue
        idx = rng.choice(cols, size=int(rng.integers(1, 300)), replace=False)
        dense[r, idx] = (rng.standard_normal(len(idx)) + 1j * rng.standard_normal(len(idx))) / cols
    ks = scipy.sparse.csr_matrix(dense)
    gs = zafx.cqtspectrogram_batch(x[:1, :20000], 8000, 50, ks)
    ref = orc.cqtspectrogram(x[0, :20000].astype(np.float64), 8000, 50, ks)
    assert gs[0].shape == ref.shape and relerr(gs[0], ref) <= TOL_FB
    assert np.all(gs[0][5] == 0)


@pytest.mark.parametrize("fft_length,rows,seed", [(32768, 145, 0), (32768, 1, 1), (16384, 7, 2), (4096, 33, 3), (1024, 64, 4), (32768, 300, 5)])
def test_cqt_matrix_core_contraction(zafx, fft_length, rows, seed):
    """k_cqt's matrix-core contraction (real matrices whose columns lie in the lower half: rows in pairs, columns in segments, two segments per
    4-lane block of v_mfma_f32_4x4x1): odd row counts (a pair of one), empty rows, rows of one entry, scattered (non-contiguous) columns,
    columns 0 / N / N/2, duplicate-free unions of very different rows in a pair, frame lengths with fewer than 16 waves -- and the same
    matrix with ONE column moved into the upper half (a conjugated bin), which must take the lane-reduction form and agree as well."""
    rng = np.random.default_rng([77, seed])
    half = fft_length // 2
    dense = np.zeros((rows, fft_length))
    for r in range(rows):
        if rows > 4 and r % 11 == 3:
            continue                                   # an empty row
        kind = r % 3
        if kind == 0:                                  # a band, as the reference's kernels
            lo = int(rng.integers(1, half - 40))
            idx = np.arange(lo, lo + int(rng.integers(1, 40)))
        elif kind == 1:                                # scattered
            idx = rng.choice(half + 1, size=int(rng.integers(1, 60)), replace=False)
        else:                                          # the special columns among others
            idx = np.unique(np.concatenate([[0, half, half // 2], rng.choice(half, size=5, replace=False)]))
        dense[r, idx] = rng.standard_normal(len(idx)) / fft_length
    fs, tr = 8000, 40
    n = 3 * fft_length + 1234
    x = np.stack([synth_clip(31, c, n) for c in range(2)])
    for variant in ("lower half", "one conjugated bin"):
        d = dense.copy()
        if variant == "one conjugated bin":
            r = 0 if rows == 1 else 1
            d[r, fft_length - 7] = 0.5 / fft_length
        k = scipy.sparse.csr_matrix(d)
        got = zafx.cqtspectrogram_batch(x, fs, tr, k)
        for c in range(2):
            ref = orc.cqtspectrogram(x[c].astype(np.float64), fs, tr, k)
            assert got[c].shape == ref.shape and relerr(got[c], ref) <= TOL_FB, (variant, c)
        if rows > 4:
            assert np.all(got[0][3] == 0), variant      # the empty row
    if rows % 12 == 0 or rows == 145:                   # chromagram through the same contraction (its own barrier behind the finishing pass)
        res = 12
        if rows % res == 0:
            k = scipy.sparse.csr_matrix(dense)
            ch = zafx.cqtchromagram_batch(x[:1], fs, tr, res, k)
            assert relerr(ch[0], orc.cqtchromagram(x[0].astype(np.float64), fs, tr, res, k)) <= TOL_FB


def test_device_buffer_placed_keeps_the_fastest_candidate(zafx):
    """DeviceBuffer.placed: every candidate is initialised and probed (the first once more, uncounted), the one with the lowest
    probe time stays usable, the others are freed."""
    seen = []

    def init(buf):
        buf.upload(np.full(buf.shape, len(seen), np.float32))

    def probe(buf):
        seen.append(buf.ptr.value)
        return {0: 9.0, 1: 9.0, 2: 3.0, 3: 1.0, 4: 2.0}[len(seen) - 1]   # (call 0 is the uncounted one)

    buf, times = zafx.DeviceBuffer.placed((4, 1000), np.float32, probe, candidates=4, init=init)
    assert times == [9.0, 3.0, 1.0, 2.0] and len(set(seen)) == 4 and seen[0] == seen[1]
    assert buf.ptr.value == seen[3]
    assert np.array_equal(buf.download(), np.full((4, 1000), 3, np.float32))   # (initialised when four probes had been made)
    buf.free()


# ------------------------------------------------------------------ round 3: chunked double-buffered host path, RCCL record
def test_run_host_chunks_over_two_lanes(zafx):
    """zafx_run_host: clips in chunks over two streams (upload / kernel / download of neighbouring chunks overlap) must give
    exactly what one chunk gives, for every chunk size (ragged last chunk, one clip per chunk, more chunks than lanes), with
    pageable and page-locked arrays, forward and inverse kinds, both layouts and padded rows."""
    x = np.stack([synth_clip(11, c, 30000 + 0) for c in range(11)])
    ham, kbd = zafx.hamming(2048), zafx.kaiser_bessel_derived(2048)
    fb = zafx.melfilterbank(44100, 2048, 128)
    plans = [zafx.stft_plan(ham, 1024), zafx.stft_plan(ham, 1024, layout="TF"), zafx.stft_plan(ham, 1024, onesided=True),
             zafx.stft_plan(ham, 1000, row_align=16), zafx.mdct_plan(kbd), zafx.mel_plan(ham, 1024, fb), zafx.mel_plan(ham, 1024, fb, 20)]
    for plan in plans:
        whole = plan.run_host(x, x.shape[1], chunk_clips=len(x))
        for chunk in (1, 2, 3, 4, 10, 0):
            assert np.array_equal(plan.run_host(x, x.shape[1], chunk_clips=chunk), whole), (plan.kernel_name, chunk)
        pin_in = zafx.pinned_empty(x.shape, np.float32)
        pin_in[:] = x
        base_shape = plan.out_shape(len(x), x.shape[1])
        pin_out = zafx.pinned_empty(base_shape, plan.out_dtype)
        got = plan.run_host(pin_in, x.shape[1], out=pin_out, chunk_clips=3)
        assert np.array_equal(got, whole)
    ref = orc.stft_batch(x.astype(np.float64), ham, 1024)
    got = zafx.stft_batch(x, ham, 1024)
    for c in range(len(x)):
        assert relerr(got[c], ref[c]) <= TOL_FFT
    # inverse kinds: the 2-D side is the input
    spec = got
    inv = zafx.istft_plan(ham, 1024)
    whole = inv.run_host(spec, spec.shape[2], chunk_clips=len(x))
    for chunk in (1, 4, 0):
        assert np.array_equal(inv.run_host(spec, spec.shape[2], chunk_clips=chunk), whole)
    assert np.max(np.abs(whole[:, :x.shape[1]] - x)) < 1e-5
    coef = zafx.mdct_batch(x, kbd)
    imd = zafx.mdct_plan(kbd, inverse=True)
    whole = imd.run_host(coef, coef.shape[2], chunk_clips=len(x))
    assert np.array_equal(imd.run_host(coef, coef.shape[2], chunk_clips=2), whole)
    # a destination of the wrong shape / dtype is refused before anything is written
    plan = plans[0]
    with pytest.raises(ValueError):
        plan.run_host(x, x.shape[1], out=np.empty((11, 2048, 5), np.complex64))
    with pytest.raises(ValueError):
        plan.run_host(x, x.shape[1], out=np.empty(plan.out_shape(11, x.shape[1]), np.complex128))
    # f64 plans share a scratch: one lane, same result for every chunking
    p64 = zafx.istft_plan(ham, 1024, f64=True)
    s64 = spec.astype(np.complex128)
    assert np.array_equal(p64.run_host(s64, s64.shape[2], chunk_clips=2), p64.run_host(s64, s64.shape[2], chunk_clips=11))


def test_comm_reports_what_rccl_saw(zafx):
    comm = zafx.Comm(0, 0, 1, zafx.Comm.unique_id())
    assert comm.count() == 1 and comm.user_rank() == 0
    comm.destroy()


def _two_rank_worker():
    """Body of one rank of test_rccl_two_ranks (run as a subprocess: one process per GPU)."""
    import hashlib
    import sys
    import zafx
    from zafx import launch
    rank, local_rank, world = launch.rank_env()
    rdzv = launch.Rendezvous.from_env(timeout=120.0)
    uid = rdzv.broadcast(zafx.Comm.unique_id() if rank == 0 else b"")
    comm = zafx.Comm(local_rank, rank, world, uid)
    assert comm.count() == world and comm.user_rank() == rank
    ham = zafx.hamming(2048)
    fb = zafx.melfilterbank(44100, 2048, 128)
    # rank 1 starts from WRONG constants: only the broadcast can make its results equal rank 0's
    plan = zafx.mel_plan(ham if rank == 0 else ham[::-1].copy(), 1024, fb, 20, device=local_rank)
    comm.broadcast_constants(plan, root=0)
    n_clips, n = 6, 30000
    lo, hi = zafx.clip_range(n_clips, rank, world)
    x = np.stack([synth_clip(21, c, n) for c in range(lo, hi)])
    got = plan.run_host(x, n)
    ref = np.stack([orc.mfcc(c.astype(np.float64), ham, 1024, fb, 20) for c in x])
    err = max(relerr(g, r) for g, r in zip(got, ref))
    parts = rdzv.all_gather(f"{lo}:{hi}:{err:.3e}:{hashlib.sha1(got.tobytes()).hexdigest()}".encode())
    comm.destroy()
    rdzv.close()
    if rank == 0:
        print("RANKS " + " ".join(p.decode() for p in parts))
    sys.exit(0 if err <= TOL_FB else 5)


def test_rccl_two_ranks(zafx):
    """The real 2-rank flow wherever two GPUs exist: file rendezvous -> ncclCommInitRank -> broadcast of window / filterbank / DCT
    rows from rank 0 over xGMI (rank 1 holds a wrong window before it) -> every rank transforms its own clip range -> parity of
    both shards against the oracle.  Skipped on one-GPU boxes."""
    import os
    import sys
    if zafx.device_count() < 2:
        pytest.skip("needs two GPUs")
    from zafx import launch
    code, out = launch.spawn_ranks(["-c", "import sys; sys.path[:0] = %r; import test_gpu_parity as t; t._two_rank_worker()"
                                    % [os.path.dirname(os.path.abspath(__file__))] + ""], 2, timeout=300)
    assert code == 0, out
    line = [ln for ln in out.splitlines() if ln.startswith("RANKS ")][-1].split()[1:]
    assert [p.split(":")[:2] for p in line] == [["0", "3"], ["3", "6"]]


@pytest.mark.parametrize("wl,hop,n,clips,onesided", [
    (2048, 1024, 442024, 5, False),   # T = 433: the bench line's off-grid geometry
    (2048, 1024, 442024, 3, True),    # one-sided: 1025 rows, every clip starts at another phase of the line grid
    (2048, 1024, 448168, 2, False),   # T = 439, two clips over 256 workgroups: segments of a few tiles each
    (2048, 1024, 30001, 4, False),    # odd clip length: the predicated loads
    (2048, 512, 100000, 3, False),    # hop W/4, T = 197
    (2048, 1554, 50000, 3, False),    # a hop that is no divisor of anything
    (512, 256, 9000, 3, False),       # smaller windows run the same kernel
    (256, 64, 5100, 7, True),
    (2048, 1024, 1000, 2, False),     # fewer frames than a tile
    (2048, 1024, 481440, 1, False),   # 10.03 s at 48 kHz, one clip: 30 segments
])
def test_stft_rows_off_the_line_grid(zafx, wl, hop, n, clips, onesided):
    """k_stft_ft16c (round 3): when T is not a multiple of 16 the rows of the compact (W, T) array straddle 128-byte lines; the
    kernel walks a clip's tiles in order and completes every line from the previous tile's values carried in registers.
    Values must not depend on the route: compared with the oracle clip by clip, first and last frames included."""
    w = zafx.hamming(wl)
    x = np.stack([synth_clip(31, c, n) for c in range(clips)])
    got = zafx.stft_batch(x, w, hop, onesided=onesided)
    assert got.shape[-1] % 16 != 0
    for c in range(clips):
        ref = orc.stft(x[c].astype(np.float64), w, hop)
        ref = ref[:wl // 2 + 1] if onesided else ref
        assert got[c].shape == ref.shape
        assert relerr(got[c], ref) <= TOL_FFT
        assert relerr(got[c][:, -3:], ref[:, -3:]) <= 1e-4 and relerr(got[c][:, :3], ref[:, :3]) <= 1e-4
    # the padded layout (other kernel, other butterfly schedule) holds the same numbers to rounding
    pad = zafx.stft_plan(w, hop, onesided=onesided, row_align=16).run_host(x, n)
    assert relerr(pad, got) <= 2e-6


@pytest.mark.parametrize("wl,n,clips", [(2048, 441000 + 2048, 3), (2048, 33000, 5), (2048, 441000 + 8192, 2), (1024, 20600, 4), (512, 9300, 6),
                                        (2048, 1024 * 471, 1), (2048, 3000, 2),
                                        (2048, 30000, 3), (1024, 20000, 2), (2048, 442024, 3)])   # (the last three: odd T, 4-byte stores)
def test_mdct_rows_off_the_line_grid(zafx, wl, n, clips):
    """k_mdct_ft32's carry form (round 3): even T that is not a multiple of 16 -- rows that are not whole half lines are
    completed from the previous tile's pairs carried in registers (odd T: the two floats of a pair choose separately).  Every clip against the oracle,
    and the inverse back."""
    w = zafx.kaiser_bessel_derived(wl)
    x = np.stack([synth_clip(41, c, n) for c in range(clips)])
    got = zafx.mdct_batch(x, w)
    T = got.shape[-1]
    assert T % 16 != 0
    for c in range(clips):
        ref = orc.mdct(x[c].astype(np.float64), w)
        assert got[c].shape == ref.shape and relerr(got[c], ref) <= TOL_FFT
        assert relerr(got[c][:, -2:], ref[:, -2:]) <= 1e-4 and relerr(got[c][:, :2], ref[:, :2]) <= 1e-4
    y = zafx.imdct_batch(got, w)
    assert np.max(np.abs(y[:, :n - 1] - x[:, :n - 1])) < 1e-5
    pad = zafx.mdct_plan(w, row_align=32).run_host(x, n)
    assert relerr(pad, got) <= 2e-6


@pytest.mark.parametrize("fmin,fmax,tr,n,clips", [(27.5, 3520.0, 25, 200000, 3),    # 65536, the low band: the float32 double form
                                                  (32.7, 1046.5, 50, 70001, 2),     # odd clip length: every frame through the predicated loads
                                                  (27.5, 880.0, 10, 66000, 9),      # more clips than groups, one or two frames each
                                                  (27.5, 8000.0, 25, 140000, 2)])   # columns above bin 8191: routed to the float64 kernel
def test_cqt_fft_length_65536(zafx, fmin, fmax, tr, n, clips):
    """fft_length 65536 (minimum frequencies below 46 Hz at 44.1 kHz, zaf.py:505-509) on the float32 kernel: the frame is transformed
    as the even and the odd bins of two 16384-point transforms (k_cqt's double form, round 3).  Against the oracle clip by clip;
    a complex-valued copy of the kernel takes the streamed complex contraction of the same form."""
    fs, res = 44100, 24
    ck = zafx.cqtkernel(fs, res, fmin, fmax)
    assert ck.shape[1] == 65536
    plan = zafx.cqt_plan(fs, tr, ck)
    low = np.minimum(ck.indices, 65536 - ck.indices).max() <= 8191
    assert plan.f64 == (not low)
    x = np.stack([synth_clip(61, c, n) for c in range(clips)])
    got = zafx.cqtspectrogram_batch(x, fs, tr, ck)
    chroma = zafx.cqtchromagram_batch(x, fs, tr, res, ck)
    assert got.dtype == np.float32
    for c in range(clips):
        x64 = x[c].astype(np.float64)
        ref = orc.cqtspectrogram(x64, fs, tr, ck)
        assert got[c].shape == ref.shape and relerr(got[c], ref) <= TOL_FB
        assert relerr(chroma[c], orc.cqtchromagram(x64, fs, tr, res, ck)) <= TOL_FB
    if low:
        ckc = ck.copy().astype(np.complex128)
        ckc.data = ckc.data * np.exp(0.3j)          # a kernel that is not numerically real: same magnitudes
        gotc = zafx.cqtspectrogram_batch(x[:1], fs, tr, scipy.sparse.csr_matrix(ckc))
        assert relerr(gotc[0], orc.cqtspectrogram(x[0].astype(np.float64), fs, tr, ck)) <= TOL_FB
