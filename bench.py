#!/usr/bin/env python3
"""Benchmark of the windowed-transform hot path.  Headline = batched STFT (BASELINE.json configs[1]) in audio
Msamples/s; the same JSON line carries every other BASELINE config under "configs".

    python bench.py [--gpus N] [--steps K] [--warmup W] [--kind all|stft|istft|mdct|imdct|mel|mfcc|cqt|...]
    <any launcher that sets RANK / LOCAL_RANK / WORLD_SIZE, one process per GPU> bench.py --gpus N ...

A "step" is one pass of the hot path over one batch: 1024 clips x 10 s @ 44.1 kHz per GPU (config 5, CQT: 1024
clips x 30 s per GPU = 8192 over 8 GPUs).  Weak scaling: every rank transforms its own clips; clips are
independent, there is no data-path collective -- the only communication is the RCCL broadcast of the window /
filterbank / CQT-kernel constants from rank 0 before the timed region.  Inputs and outputs are resident in HBM
when the timed region starts.  Rank 0 prints ONE JSON line.

N > 1 without a framework (zafx/launch.py): under a launcher that sets RANK / LOCAL_RANK / WORLD_SIZE (the driver's
`python -m torch.distributed.run` does; this script never imports torch) the ranks meet through a file rendezvous; with
WORLD_SIZE unset, `--gpus N` starts the N ranks itself.  With more than one rank the RCCL communicator is part of the
record: the line carries what RCCL itself reports (`rccl.ranks_seen`, per-rank kernel times, broadcast wall time), and a
run whose ranks cannot form it prints its line with an "error" field and exits 3 (ZAFX_BENCH_ALLOW_NO_COMM=1: exit 0).  Test aids (environment): ZAFX_BENCH_FORCE_DIST=1 runs the
N-rank plumbing (rendezvous, communicator, broadcasts) with one rank; ZAFX_BENCH_SHARE_DEVICES=1 lets N ranks share fewer GPUs
(rank r on GPU r mod count; RCCL then refuses the duplicate device, so this aid needs ZAFX_BENCH_ALLOW_NO_COMM=1); ZAFX_BENCH_COMM_TIMEOUT
(seconds, default 180) bounds the wait for the RCCL communicator.

Timing: the device is first kept busy with the plan for ZAFX_BENCH_PREWARM_S seconds (default 0.4, untimed, whatever
--warmup says: the clocks of an idle MI355X ramp over the first few hundred milliseconds of work, so a short warm-up count
leaves a ~1 ms kernel inside the ramp), then the contract's W warm-up steps and K timed steps follow.  `value` comes from the
FIRST allocation of every buffer, as a caller of the library gets it; where the 7 GB spectrum lands in physical memory moves
the STFT by up to 12 %, so the line also reports a survey of further allocations (`config.placement`).

`roofline.achieved` = algorithmic bytes (or flops) per launch / mean kernel duration measured with HIP events on
the plan's stream over the timed region.  `cpu_baseline` = the NumPy oracle (a port of zaf.py, same NumPy calls per
clip) timed on this host, rank 0, N = 1 only, bounded samples.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "zaf-python_amd"))
sys.path.insert(0, ROOT)

FS, W, H = 44100, 2048, 1024
HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6290.0   # same guide: measured-achievable copy rate
F32_PEAK_TFLOPS = 157.3    # dense f32 vector / f32-input MFMA peak
F64_PEAK_TFLOPS = 78.6     # float64 vector: half the f32 vector rate (v_fma_f64 issues every 4 cycles per wave: 256 CUs x 4 SIMDs x 32 FLOP/clk x 2.4 GHz; the guide lists no f64 row)
PLACEMENT_SURVEY = 6       # further allocations of the headline's row-strided output probed AFTER every timed region (reported, never timed)
CONFIG_KINDS = ("istft", "mel", "mfcc", "mel_mfcc", "mdct", "imdct", "cqt")   # SURVEY 8(a) a2 + BASELINE configs 3 (mel_mfcc: in one pass), 4, 5 (config 2 = the headline)
EXTRA_KINDS = ("stft1", "stftmag", "istft1", "stft_offgrid", "mdct_offgrid", "stft4096", "stft4096_h1024", "istft4096", "mdct4096", "mel4096", "dct", "stft64", "mdct64", "istft_offgrid", "imdct_offgrid", "stftmag_offgrid", "stft8192", "mdct8192", "istft64", "imdct64", "mel64", "mfcc64", "cqt64", "istft_offgrid_compact", "stftmag_offgrid_compact", "mel_pcm16", "mdct_pcm16", "dct1000", "istft8192", "imdct8192", "stft_offgrid_padded", "mdct_offgrid_padded", "stft4096_padded", "istft4096_padded", "mdct4096_padded")   # one-sided pair (8f rank 4) and geometries off the benchmark's grid


def synth(seed, c, n):
    return np.random.default_rng([seed, c]).standard_normal(n).astype(np.float32)


def issued_mfma_flops_per_tile(dense, frames_per_tile=16):
    """f32 MFMA work k_mel really issues for one 16-frame tile: per 16-row block only the K-steps (4 columns each) of
    the band that holds non-zeros (pack_band in zafx_capi.cpp) -- not the dense-equivalent 2*rows*cols."""
    steps = 0
    for b in range(0, dense.shape[0], 16):
        cols = np.nonzero(np.any(dense[b:b + 16] != 0, axis=0))[0]
        if len(cols):
            steps += (cols[-1] - (cols[0] & ~3)) // 4 + 1
    return steps, steps * 2.0 * 16 * 4 * frames_per_tile


def replicate64(base, n_clips, device):
    """The distinct clips as float64, replicated on the device to n_clips."""
    import zafx
    d_b = zafx.DeviceBuffer.from_host(base.astype(np.float64), device)
    d_x = zafx.DeviceBuffer((n_clips, base.shape[1]), np.float64, device)
    for r in range(n_clips // base.shape[0]):
        d_x.copy_from(d_b, dst_offset=r * base.nbytes * 2)
    d_b.free()
    return d_x


class PcmPlan:
    """A plan fed with int16 PCM that is already on the device (zafx_execute_pcm: wavread's x / 2^15, zaf.py:1202, inside the kernel's own loads):
    execute() of the timing loops = Plan.execute_pcm; everything else is the plan's."""

    def __init__(self, plan, channels=1):
        self._plan, self._channels = plan, channels

    def execute(self, d_in, d_out, n_clips, n_in):
        self._plan.execute_pcm(d_in, d_out, n_clips, n_in, self._channels)

    def __getattr__(self, name):
        return getattr(self._plan, name)


def make_workload(kind, device, layout="FT"):
    """dict(plan, n_clips, n_in, d_in, d_out, samples_per_clip, bytes_per_launch, valu_flops, mfma_flops, desc, ...)."""
    import zafx
    ham = zafx.hamming(W)
    kbd = zafx.kaiser_bessel_derived(W)
    B, N = 1024, 441000
    distinct = 8
    T = 432
    if kind == "cqt":
        B, N, T = 1024, 1323000, 750   # BASELINE config 5: 8192 clips x 30 s over 8 GPUs = 1024 per GPU
    if kind == "dct":
        B, N, T = 16384, 1024, 1
    if kind == "dct1000":
        B, N, T = 16384, 1000, 1   # a length off the power-of-two grid: the chirp-z form (k_dct_bs32)
    if kind == "cqt64":
        B, N, T = 256, 1323000, 750   # float64 clips of 30 s are 10.6 MB: a quarter of a GPU's share of config 5 (2.7 GB), same frames per clip
    if kind in ("stft64", "mdct64", "istft64", "imdct64", "mel64", "mfcc64"):
        B = 1024                  # the reference's own dtype (zaf.py:128, :139: float64 in, complex128 out): 40.1 / 16 B per sample
    padded_rows = kind.endswith("_padded")    # stft / mdct off the grid on row-padded device arrays (what stft_batch / mdct_batch do by default since round 6)
    kind = kind[:-len("_padded")] if padded_rows else kind
    compact_rows = kind.endswith("_compact")   # istft / stftmag off the grid: the *_batch functions' default (rows padded to 128-byte lines, round 6) and, `_compact`, the reference's own memory order
    kind = kind[:-len("_compact")] if compact_rows else kind
    if kind in ("stft_offgrid", "istft_offgrid", "stftmag_offgrid"):
        N, T = 442024, 433        # one more frame than config 2: rows of 433 complex64 = 3464 B, off the 128-byte grid
    if kind == "stft4096":
        T = 217
    if kind in ("mdct_offgrid", "imdct_offgrid"):
        N, T = 442024, 433       # ceil(N / 1024) + 1 (zaf.py:1033): float32 rows of 1732 B
    if kind in ("stft8192", "istft8192"):
        N, T = 4096 * 111, 112    # W = 8192, hop 4096, on the line grid: ceil(N / 4096) + 1 frames
    if kind in ("mdct8192", "imdct8192"):
        N, T = 4096 * 127, 128    # W = 8192 MDCT on the line grid: ceil(N / 4096) + 1 frames (zaf.py:1033)
    if kind in ("mdct4096", "mel4096", "istft4096"):
        T = 217                   # mdct: ceil(N / 2048) + 1 (odd: rows off the line grid); mel: hop 2048
    base = np.stack([synth(0, c, N) for c in range(distinct)])
    d_base = zafx.DeviceBuffer.from_host(base, device)
    d_x = zafx.DeviceBuffer((B, N), np.float32, device)
    for r in range(B // distinct):
        d_x.copy_from(d_base, dst_offset=r * distinct * N * 4)
    d_base.free()
    wl = dict(kind=kind + ("_compact" if compact_rows else "_padded" if padded_rows else ""), n_clips=B, samples_per_clip=N, base=base, valu_flops=0.0, valu_flops_real_input=0.0, mfma_flops=0.0, flops_note=None, frames=T)
    pad_c64 = 16 if kind == "istft_offgrid" and not compact_rows else 0      # rows of 433 complex64 -> pitch 448
    pad_f32 = 32 if kind == "stftmag_offgrid" and not compact_rows else 0    # rows of 433 float32 -> pitch 448
    if kind == "stft":
        plan = zafx.stft_plan(ham, H, layout=layout, device=device)
        wl.update(plan=plan, d_in=d_x, n_in=N, bytes_per_launch=B * (4 * N + 8 * W * T),
                  desc="Batched STFT: 1024 clips x 10 s @ 44.1 kHz, Hamming win=2048 hop=1024, two-sided c64 (W,T) layout")
    elif kind == "stft_offgrid":   # the compact reference layout when T is not a multiple of 16
        plan = zafx.stft_plan(ham, H, layout=layout, device=device, row_align=16 if padded_rows else 0)
        wl.update(plan=plan, d_in=d_x, n_in=N, bytes_per_launch=B * (4 * N + 8 * W * T),
                  desc="Batched STFT off the line grid: 1024 clips x 442024 samples, win=2048 hop=1024, T = 433" + (", device rows padded to 128-byte lines (pitch 448: stft_batch's "
                       "default off the grid; algorithmic bytes of the compact array)" if padded_rows else " (rows straddle 128-byte lines), compact (W,T) layout"))
    elif kind == "stft4096":
        plan = zafx.stft_plan(zafx.hamming(4096), 2048, layout=layout, device=device, row_align=16 if padded_rows else 0)
        wl.update(plan=plan, d_in=d_x, n_in=N, bytes_per_launch=B * (4 * N + 8 * 4096 * T),
                  desc="Batched STFT, win=4096 hop=2048: 1024 clips x 10 s, T = 217, two-sided c64 (W,T) layout" + (", device rows padded to 128-byte lines (pitch 224: stft_batch's default off the grid; "
                       "algorithmic bytes of the compact array)" if padded_rows else ""))
    elif kind == "stft4096_h1024":   # W = 4096 on the line grid (T = 432): k_stft_ft16b, two bands of bins per 16-frame tile
        plan = zafx.stft_plan(zafx.hamming(4096), 1024, layout=layout, device=device)
        wl.update(plan=plan, d_in=d_x, n_in=N, bytes_per_launch=B * (4 * N + 8 * 4096 * T),
                  desc="Batched STFT, win=4096 hop=1024: 1024 clips x 10 s, T = 432, two-sided c64 (W,T) layout")
    elif kind == "istft4096":        # k_istft_ft16b: one band of samples (even / odd packed samples) per workgroup; T = 217 is odd: 8-byte row pieces
        fwd = zafx.stft_plan(zafx.hamming(4096), 2048, device=device, row_align=16 if padded_rows else 0)
        d_s = zafx.DeviceBuffer(fwd.out_shape(B, N), np.complex64, device)
        fwd.execute(d_x, d_s, B, N)
        INNER_LOG.append({"kind": None, "kernel": fwd.last_kernel, "launches": 1})
        fwd.sync()
        d_x.free()
        plan = zafx.istft_plan(zafx.hamming(4096), 2048, device=device, row_align=16 if padded_rows else 0)
        wl.update(plan=plan, d_in=d_s, n_in=T, bytes_per_launch=B * (8 * 4096 * T + 4 * (T * 2048 - 2048)),
                  desc="Batched ISTFT, win=4096 hop=2048: 1024 clips x 217 frames" + (" (device rows padded to 128-byte lines: istft_batch's default off the grid)" if padded_rows else ""))
    elif kind == "stft8192":         # k_stft_ft16q: four classes of bins per 16-frame tile
        plan = zafx.stft_plan(zafx.hamming(8192), 4096, layout=layout, device=device)
        wl.update(plan=plan, d_in=d_x, n_in=N, bytes_per_launch=B * (4 * N + 8 * 8192 * T),
                  desc="Batched STFT, win=8192 hop=4096: 1024 clips x 10.3 s, T = 112, two-sided c64 (W,T) layout")
    elif kind in ("istft8192", "imdct8192"):   # the inverse kinds at W = 8192 (round 6: k_istft_ft8q), input = the device's own forward result
        w8 = zafx.hamming(8192) if kind == "istft8192" else zafx.kaiser_bessel_derived(8192)
        fwd = zafx.stft_plan(w8, 4096, device=device) if kind == "istft8192" else zafx.mdct_plan(w8, device=device)
        d_s = zafx.DeviceBuffer(fwd.out_shape(B, N), fwd.out_dtype, device)
        fwd.execute(d_x, d_s, B, N)
        INNER_LOG.append({"kind": None, "kernel": fwd.last_kernel, "launches": 1})
        fwd.sync()
        d_x.free()
        if kind == "istft8192":
            plan = zafx.istft_plan(w8, 4096, device=device)
            wl.update(plan=plan, d_in=d_s, n_in=T, bytes_per_launch=B * (8 * 8192 * T + 4 * (T - 1) * 4096),
                      desc="Batched ISTFT, win=8192 hop=4096: 1024 clips x 112 frames, two-sided c64 (W,T) layout")
        else:
            plan = zafx.mdct_plan(w8, device=device, inverse=True)
            wl.update(plan=plan, d_in=d_s, n_in=T, bytes_per_launch=B * (4 * 4096 * T + 4 * (4096 * (T - 1) - 1)),
                      desc="Batched IMDCT, KBD win=8192: 1024 clips x 128 frames")
    elif kind == "mdct8192":         # k_mdct_ft32q: four bands of bins per 32-frame tile
        plan = zafx.mdct_plan(zafx.kaiser_bessel_derived(8192), device=device)
        wl.update(plan=plan, d_in=d_x, n_in=N, bytes_per_launch=B * (4 * N + 4 * 4096 * T),
                  desc="Batched MDCT, KBD win=8192: 1024 clips x 11.8 s, T = 128, compact (W/2,T) layout")
    elif kind == "mdct4096":         # k_mdct_ft32b (32-frame tiles, two bands of bins); T = 217 is odd: rows off the line grid
        plan = zafx.mdct_plan(zafx.kaiser_bessel_derived(4096), device=device, row_align=32 if padded_rows else 0)
        wl.update(plan=plan, d_in=d_x, n_in=N, bytes_per_launch=B * (4 * N + 4 * 2048 * T),
                  desc="Batched MDCT, KBD win=4096: 1024 clips x 10 s, T = 217, " + ("device rows padded to 128-byte lines (pitch 224: mdct_batch's default off the grid; algorithmic bytes of the "
                       "compact array)" if padded_rows else "compact (W/2,T) layout"))
    elif kind == "mel4096":          # k_mel_ft16b: the two-band STFT kernel with the filterbank product in place of the stores
        plan = zafx.mel_plan(zafx.hamming(4096), 2048, zafx.melfilterbank(FS, 4096, 128), device=device)
        wl.update(plan=plan, d_in=d_x, n_in=N, bytes_per_launch=B * (4 * N + 4 * 128 * T),
                  desc="melspectrogram, win=4096 hop=2048, 128 filters: 1024 clips x 10 s (fused on the two-band kernel, float32)")
    elif kind == "stft64":  # float64 device arithmetic (written for exactness, not speed)
        d_x.free()
        d_x = replicate64(base, B, device)
        plan = zafx.stft_plan(ham, H, layout=layout, device=device, f64=True)
        wl.update(plan=plan, d_in=d_x, n_in=N, bytes_per_launch=B * (8 * N + 16 * W * T),
                  desc="Batched STFT in float64 / complex128 (the reference's dtype): 1024 clips x 10 s, Hamming win=2048 hop=1024, two-sided")
    elif kind == "mdct64":
        d_x.free()
        d_x = replicate64(base, B, device)
        plan = zafx.mdct_plan(kbd, device=device, f64=True)
        wl.update(plan=plan, d_in=d_x, n_in=N, bytes_per_launch=B * (8 * N + 8 * (W // 2) * T),
                  desc="Batched MDCT in float64: 1024 clips x 10 s, KBD win=2048")
    elif kind in ("istft64", "imdct64"):   # the float64 inverse pair on the tiled structure (k_istft_ft8_f64, k_imdct_ft16_f64), input = the device's own forward result
        d_x.free()
        d_x = replicate64(base, B, device)
        fwd = zafx.stft_plan(ham, H, device=device, f64=True) if kind == "istft64" else zafx.mdct_plan(kbd, device=device, f64=True)
        d_s = zafx.DeviceBuffer(fwd.out_shape(B, N), fwd.out_dtype, device)
        fwd.execute(d_x, d_s, B, N)
        INNER_LOG.append({"kind": None, "kernel": fwd.last_kernel, "launches": 1})
        fwd.sync()
        d_x.free()
        if kind == "istft64":
            plan = zafx.istft_plan(ham, H, device=device, f64=True)
            wl.update(plan=plan, d_in=d_s, n_in=T, bytes_per_launch=B * (16 * W * T + 8 * (T * H - (W - H))),
                      desc="Batched ISTFT in float64 of the device STFT of the same batch: 1024 clips x 432 frames, Hamming win=2048 hop=1024")
        else:
            plan = zafx.mdct_plan(kbd, device=device, inverse=True, f64=True)
            wl.update(plan=plan, d_in=d_s, n_in=T, bytes_per_launch=B * (8 * (W // 2) * T + 8 * ((W // 2) * (T - 1) - 1)),
                      desc="Batched IMDCT in float64 of the device MDCT of the same batch: 1024 clips x 432 frames, KBD win=2048")
    elif kind in ("mel64", "mfcc64"):   # melspectrogram / mfcc in the reference's dtype on the tiled structure (k_mel_ft8_f64)
        d_x.free()
        d_x = replicate64(base, B, device)
        fb = zafx.melfilterbank(FS, W, 128)
        rows = 128 if kind == "mel64" else 20
        plan = zafx.mel_plan(ham, H, fb, None if kind == "mel64" else 20, device=device, f64=True)
        wl.update(plan=plan, d_in=d_x, n_in=N, bytes_per_launch=B * (8 * N + 8 * rows * T),
                  desc=f"Fused {kind[:-2]} in float64: 1024 clips x 10 s, win=2048 hop=1024, 128 mel filters" + (", 20 coefficients" if kind == "mfcc64" else ""))
    elif kind == "cqt64":   # k_cqt_ft_f64
        d_x.free()
        d_x = replicate64(base, B, device)
        plan = zafx.cqt_plan(FS, 25, zafx.cqtkernel(FS, 24, 55, 3520), device=device, f64=True)
        wl.update(plan=plan, d_in=d_x, n_in=N, bytes_per_launch=B * (8 * N + 8 * 144 * T),
                  valu_flops=B * T * 5.0 * 32768 * 15, valu_flops_real_input=B * T * 5.0 * 16384 * 14,
                  desc="cqtspectrogram in float64: 256 clips x 30 s @ 44.1 kHz, 24 bins/octave 55-3520 Hz, 25 frames/s")
    elif kind == "stft1":   # one-sided output (rows 0..W/2), not the headline
        plan = zafx.stft_plan(ham, H, layout=layout, device=device, onesided=True)
        wl.update(plan=plan, d_in=d_x, n_in=N, bytes_per_launch=B * (4 * N + 8 * (W // 2 + 1) * T),
                  desc="Batched STFT, one-sided output (W/2+1, T): 1024 clips x 10 s, Hamming win=2048 hop=1024")
    elif kind in ("stftmag", "stftmag_offgrid"):   # the spectrogram the reference's examples compute (zaf.py:83), |X| of rows 0..W/2 as float32 (k_mel2, MODE 2)
        plan = zafx.stft_plan(ham, H, layout=layout, device=device, onesided="magnitude", row_align=pad_f32)
        wl.update(plan=plan, d_in=d_x, n_in=N, bytes_per_launch=B * (4 * N + 4 * (W // 2 + 1) * T),
                  desc=f"Batched magnitude spectrogram |X| (W/2+1, T) float32: 1024 clips x {T} frames, Hamming win=2048 hop=1024"
                       + (", device rows padded to 128-byte lines (the *_batch functions' default off the grid; algorithmic bytes of the compact array)" if pad_f32 else
                          ", compact rows off the 128-byte grid" if T % 32 else ""))
    elif kind == "istft1":
        fwd = zafx.stft_plan(ham, H, device=device, onesided=True)
        d_s = zafx.DeviceBuffer(fwd.out_shape(B, N), np.complex64, device)
        fwd.execute(d_x, d_s, B, N)
        INNER_LOG.append({"kind": None, "kernel": fwd.last_kernel, "launches": 1})
        fwd.sync()
        d_x.free()
        plan = zafx.istft_plan(ham, H, device=device, onesided=True)
        wl.update(plan=plan, d_in=d_s, n_in=T, bytes_per_launch=B * (8 * (W // 2 + 1) * T + 4 * (T * H - (W - H))),
                  desc="Batched ISTFT from one-sided spectra: 1024 clips x 432 frames, win=2048 hop=1024")
    elif kind in ("istft", "istft_offgrid"):   # (off the grid: T = 433, rows of 3464 bytes)
        fwd = zafx.stft_plan(ham, H, device=device, row_align=pad_c64)
        d_s = zafx.DeviceBuffer(fwd.out_shape(B, N), np.complex64, device)
        fwd.execute(d_x, d_s, B, N)
        INNER_LOG.append({"kind": None, "kernel": fwd.last_kernel, "launches": 1})
        fwd.sync()
        d_x.free()
        plan = zafx.istft_plan(ham, H, device=device, row_align=pad_c64)
        wl.update(plan=plan, d_in=d_s, n_in=T, bytes_per_launch=B * (8 * W * T + 4 * (T * H - (W - H))),
                  desc=f"Batched ISTFT: 1024 clips x {T} frames, win=2048 hop=1024" + (" (device rows padded to 128-byte lines: the *_batch functions' default off the grid; "
                       "algorithmic bytes of the compact array)" if pad_c64 else " (compact rows off the 128-byte grid)" if T % 16 else ""))
    elif kind == "mdct":
        plan = zafx.mdct_plan(kbd, device=device)
        wl.update(plan=plan, d_in=d_x, n_in=N, bytes_per_launch=B * (4 * N + 4 * (W // 2) * T),
                  desc="Batched MDCT: 1024 clips x 10 s, KBD win=2048")
    elif kind == "mdct_offgrid":   # the compact (W/2, T) layout when T is not a multiple of 16 (k_mdct_ft32's carry form)
        plan = zafx.mdct_plan(kbd, device=device, row_align=32 if padded_rows else 0)
        wl.update(plan=plan, d_in=d_x, n_in=N, bytes_per_launch=B * (4 * N + 4 * (W // 2) * T),
                  desc="Batched MDCT off the line grid: 1024 clips x 442024 samples, KBD win=2048, T = 433" + (", device rows padded to 128-byte lines (pitch 448: mdct_batch's "
                       "default off the grid; algorithmic bytes of the compact array)" if padded_rows else ", compact (W/2,T) layout"))
    elif kind in ("imdct", "imdct_offgrid"):
        fwd = zafx.mdct_plan(kbd, device=device)
        d_m = zafx.DeviceBuffer(fwd.out_shape(B, N), np.float32, device)
        fwd.execute(d_x, d_m, B, N)
        INNER_LOG.append({"kind": None, "kernel": fwd.last_kernel, "launches": 1})
        fwd.sync()
        d_x.free()
        plan = zafx.mdct_plan(kbd, device=device, inverse=True)
        wl.update(plan=plan, d_in=d_m, n_in=T, bytes_per_launch=B * (4 * (W // 2) * T + 4 * ((W // 2) * (T - 1) - 1)),
                  desc=f"Batched IMDCT of the device MDCT of the same batch: 1024 clips x {T} frames, KBD win=2048" + (" (rows off the 128-byte grid)" if T % 32 else ""))
    elif kind in ("mel", "mfcc", "mel_mfcc"):   # mel_mfcc: BASELINE config 3 as written ("melspectrogram + mfcc: same batch") from ONE set of transforms
        fb = zafx.melfilterbank(FS, W, 128)
        rows = 128 if kind == "mel" else 20 if kind == "mfcc" else 148
        plan = zafx.mel_plan(ham, H, fb, None if kind == "mel" else 20, device=device, also_mel=kind == "mel_mfcc")
        tiles = B * ((T + 15) // 16)
        # the arithmetic the kernel's real-input form executes per frame on the vector pipe: one W/2-point complex transform
        # (5 (W/2) log2(W/2)), the window (W), the split of the packed spectrum (8 per bin) and the levels (3 per bin)
        valu = B * T * (5.0 * (W // 2) * 10 + W + 11.0 * (W // 2))
        steps, per_tile = issued_mfma_flops_per_tile(fb.toarray())
        mfma = per_tile * tiles * (2 if kind == "mel_mfcc" else 1)   # (one pass: every K-step issues two matrix instructions, |X|^2 and |X|)
        steps *= 2 if kind == "mel_mfcc" else 1
        if kind != "mel":
            dsteps, dper = issued_mfma_flops_per_tile(zafx.dct2_rows(128, 20))
            steps, mfma = steps + dsteps, mfma + dper * tiles
        wl.update(plan=plan, d_in=d_x, n_in=N, bytes_per_launch=B * (4 * N + 4 * rows * T), valu_flops=valu, mfma_flops=mfma,
                  flops_note=f"vector pipe: W/2-point complex FFT + window + split + levels per frame = {valu / 1e9:.1f} GFLOP (the complex form the "
                             f"reference runs, 5 W log2 W: {B * T * 5.0 * W * 11 / 1e9:.1f}); matrix cores: {steps} issued 16x16x4 f32 K-steps per 16-frame "
                             f"tile = {mfma / 1e9:.1f} GFLOP (a dense filterbank GEMM would be {2.0 * 128 * 1024 * T * B / 1e9:.0f})",
                  desc=(f"Fused {kind}: 1024 clips x 10 s, win=2048 hop=1024, 128 mel filters" + (", 20 coefficients" if kind == "mfcc" else "")) if kind != "mel_mfcc" else
                       "melspectrogram + mfcc of the same batch in ONE pass (one set of transforms, rows 0..127 mel, 128..147 mfcc): 1024 clips x 10 s, win=2048 hop=1024, 128 mel filters, 20 coefficients")
    elif kind == "cqt":
        ck = zafx.cqtkernel(FS, 24, 55, 3520)
        plan = zafx.cqt_plan(FS, 25, ck, device=device)
        wl.update(plan=plan, d_in=d_x, n_in=N, bytes_per_launch=B * (4 * N + 4 * 144 * T),
                  valu_flops=B * T * 5.0 * 32768 * 15, valu_flops_real_input=B * T * 5.0 * 16384 * 14,
                  flops_note="SURVEY 8(d): 750 x 5*32768*15 per clip, the 32768-point complex FFT the reference runs; the real-input form the kernel "
                             "runs (one 16384-point complex transform per frame) needs 5*16384*14, under half",
                  desc="cqtspectrogram: 1024 clips x 30 s @ 44.1 kHz per GPU (config 5: 8192 clips over 8 GPUs), 24 bins/octave 55-3520 Hz, 25 frames/s")
    elif kind in ("mel_pcm16", "mdct_pcm16"):   # int16 mono in the kernel's own loads, device resident: 2 bytes per sample read
        pcm = np.clip(np.rint(base * 8192.0), -32768, 32767).astype(np.int16)   # (the same clips at -12 dBFS, as a 16-bit recorder holds them)
        d_base = zafx.DeviceBuffer.from_host(pcm, device)
        d_x.free()
        d_x = zafx.DeviceBuffer((B, N), np.int16, device)
        for r in range(B // distinct):
            d_x.copy_from(d_base, dst_offset=r * distinct * N * 2)
        d_base.free()
        wl["base"] = pcm.astype(np.float32) / 32768.0   # what the parity probe's reference starts from (zaf.py:1202)
        if kind == "mel_pcm16":
            plan = PcmPlan(zafx.mel_plan(ham, H, zafx.melfilterbank(FS, W, 128), None, device=device))
            wl.update(plan=plan, d_in=d_x, n_in=N, bytes_per_launch=B * (2 * N + 4 * 128 * T),
                      desc="Fused melspectrogram from int16 mono PCM on the device (wavread's scaling in the loads): 1024 clips x 10 s, win=2048 hop=1024, 128 mel filters")
        else:
            plan = PcmPlan(zafx.mdct_plan(kbd, device=device))
            wl.update(plan=plan, d_in=d_x, n_in=N, bytes_per_launch=B * (2 * N + 4 * (W // 2) * T),
                      desc="Batched MDCT from int16 mono PCM on the device: 1024 clips x 10 s, KBD win=2048")
    elif kind == "dct1000":   # every length the reference takes: N = 1000 = k_dct's maps around a 500-point Bluestein convolution (two 1024-point transforms per vector)
        plan = zafx.dct_plan(N, 2, device=device)
        wl.update(plan=plan, d_in=d_x, n_in=N, bytes_per_launch=B * 8 * N,
                  desc="zaf.dct type 2 of 16384 vectors x 1000 samples (a length off the power-of-two grid: N/2-point Bluestein convolution inside k_dct's maps)")
    elif kind == "dct":   # zaf.dct type 2 on the FFT core (k_dct): 8 bytes per sample, HBM-bound
        plan = zafx.dct_plan(N, 2, device=device)
        wl.update(plan=plan, d_in=d_x, n_in=N, bytes_per_launch=B * 8 * N,
                  desc="zaf.dct type 2 of 16384 vectors x 1024 samples, one 512-point complex FFT per vector (SURVEY 8f rank 3)")
    else:
        raise SystemExit(f"unknown --kind {kind}")
    # `value` and every config time come from the FIRST allocation of each buffer -- what a caller of the library gets.  (Where a
    # 7 GB row-strided array lands in physical memory moves the STFT by up to 12 %: main() surveys further allocations after
    # all timed regions and reports them under config.placement.  ZAFX_BENCH_PLACEMENT=n > 1 times the best of n instead --
    # an experiment switch, labelled in the line.)
    out_shape = plan.out_shape(B, wl["n_in"])
    n_cand = max(int(os.environ.get("ZAFX_BENCH_PLACEMENT", "1")), 1)
    if kind in ("stft", "stft1", "mdct") and n_cand > 1:
        wl["d_out"], times = zafx.DeviceBuffer.placed(out_shape, plan.out_dtype, lambda buf: probe_ms(plan, wl["d_in"], buf, B, wl["n_in"]), n_cand, device)
        wl["placement"] = {"buffer": "output", "candidates": n_cand, "probe_ms": [round(t, 4) for t in times],
                           "note": "ZAFX_BENCH_PLACEMENT: the timed buffer is the best of these allocations, NOT the first"}
        time.sleep(float(os.environ.get("ZAFX_BENCH_SETTLE_S", "0.5")))   # (the driver wipes the freed candidates in the background)
    else:
        wl["d_out"] = zafx.DeviceBuffer(out_shape, plan.out_dtype, device)
    return wl


def probe_ms(plan, d_in, d_out, n_clips, n_in, reps=8):
    for _ in range(3):
        plan.execute(d_in, d_out, n_clips, n_in)
    plan.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        plan.execute(d_in, d_out, n_clips, n_in)
    plan.sync()
    return (time.perf_counter() - t0) / reps * 1e3


def placement_survey(device, first_ms, layout="FT"):
    """After every timed region: the headline STFT into PLACEMENT_SURVEY further allocations of its 7.25 GB output, all held at
    once (a freed one would come straight back), 8 launches each.  Reported beside the first allocation's time so that a
    reader sees the spread of this box; nothing here is `value`."""
    import zafx
    n = int(os.environ.get("ZAFX_BENCH_PLACEMENT_SURVEY", PLACEMENT_SURVEY))
    if n < 1:
        return None
    wl = make_workload("stft", device, layout)
    plan, B, N = wl["plan"], wl["n_clips"], wl["n_in"]
    held, times = [wl["d_out"]], []
    try:
        prewarm(plan, wl["d_in"], wl["d_out"], B, N)
        times.append(probe_ms(plan, wl["d_in"], wl["d_out"], B, N))
        for _ in range(n):
            try:
                held.append(zafx.DeviceBuffer(held[0].shape, held[0].dtype, device))
            except zafx.ZafxError:
                break
            times.append(probe_ms(plan, wl["d_in"], held[-1], B, N))
    finally:
        for b in held[1:]:
            b.free()
        free_workload(wl)
    return {"buffer": "output (1024, 2048, 432) complex64, 7.25 GB", "timed_allocation": "first",
            "timed_kernel_ms": round(first_ms, 4), "survey_probe_ms": [round(t, 4) for t in times],
            "survey_best_ms": round(min(times), 4), "survey_worst_ms": round(max(times), 4),
            "note": "fresh allocations of the same buffer probed after all timed regions (8 launches each, host clock); "
                    "zafx.DeviceBuffer.placed picks by such a probe for a long-lived buffer; see DESIGN.md 3"}


def free_workload(wl):
    for key in ("d_in", "d_out"):
        wl[key].free()


# ---------------------------------------------------------------------------------------------------------------
# CPU side: the oracle as baseline and as checker (the only places bench.py touches oracle/)
# ---------------------------------------------------------------------------------------------------------------
def _oracle_call(kind):
    """(callable(clip index) -> result, samples per call, zaf.py lines) of the reference-faithful per-clip NumPy path."""
    from oracle import zaf_oracle as orc   # checker / baseline only
    ham, kbd = orc.hamming_periodic(W), orc.kbd_window(W)
    n = 1323000 if kind == "cqt" else 441000
    clips = [synth(0, c, n).astype(np.float64) for c in range(2)]
    if kind == "stft":
        return (lambda i: orc.stft(clips[i], ham, H)), n, "zaf.py:95-141"
    if kind == "istft":
        spec = [orc.stft(c, ham, H) for c in clips]
        return (lambda i: orc.istft(spec[i], ham, H)), n, "zaf.py:214-241"
    if kind in ("mel", "mfcc", "mel_mfcc"):
        fb = orc.melfilterbank(FS, W, 128)
        if kind == "mel_mfcc":   # what a zaf.py user runs for both: the two functions, one after the other (each with its own stft)
            return (lambda i: (orc.melspectrogram(clips[i], ham, H, fb), orc.mfcc(clips[i], ham, H, fb, 20))), n, "zaf.py:369-373 + :436-452"
        if kind == "mel":
            return (lambda i: orc.melspectrogram(clips[i], ham, H, fb)), n, "zaf.py:369-373"
        return (lambda i: orc.mfcc(clips[i], ham, H, fb, 20)), n, "zaf.py:436-452"
    if kind == "mdct":
        return (lambda i: orc.mdct(clips[i], kbd)), n, "zaf.py:1029-1073"
    if kind == "imdct":
        coef = [orc.mdct(c, kbd) for c in clips]
        return (lambda i: orc.imdct(coef[i], kbd)), n, "zaf.py:1125-1182"
    if kind == "cqt":
        ck = orc.cqtkernel(FS, 24, 55, 3520)
        return (lambda i: orc.cqtspectrogram(clips[i], FS, 25, ck)), n, "zaf.py:603-633"
    return None, 0, ""


def cpu_baseline(kind="stft", budget_s=10.0, min_calls=5):
    """The per-clip NumPy path of zaf.py (oracle port, same NumPy calls), one clip per call, 1 core."""
    call, n, lines = _oracle_call(kind)
    if call is None:
        return None
    call(0)
    call(1)   # warm-up (first calls pay FFT plans + page faults)
    times = []
    t_end = time.perf_counter() + budget_s
    i = 0
    while time.perf_counter() < t_end or len(times) < min_calls:
        t0 = time.perf_counter()
        call(i % 2)
        times.append(time.perf_counter() - t0)
        i += 1
    med = float(np.median(times))
    return {
        "value": round(n / med / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
        "sample": f"{len(times)} calls of the NumPy oracle {kind} ({lines} restated) on one {n / FS:.0f} s clip each, "
                  f"median {med * 1e3:.2f} ms/clip, min {min(times) * 1e3:.2f} ms; numpy {np.__version__}; "
                  f"host has {os.cpu_count()} logical cores, FFT single-threaded",
    }


def _pool_worker(args):
    seed, seconds = args
    from oracle import zaf_oracle as orc
    ham = orc.hamming_periodic(W)
    x = synth(0, seed % 8, 441000).astype(np.float64)
    orc.stft(x, ham, H)   # warm-up
    count, t_end = 0, time.perf_counter() + seconds
    while time.perf_counter() < t_end:
        orc.stft(x, ham, H)
        count += 1
    return count


def usable_cores():
    """Cores this process may run on (affinity mask, capped by a cgroup CPU quota when there is one)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_pool_main(n_workers, seconds):
    """All-cores figure (SURVEY 8(d) CPU baseline plan): a process pool over the host's cores, BLAS pinned to 1 thread;
    every worker transforms 10 s clips for `seconds` seconds."""
    import multiprocessing as mp
    from oracle import zaf_oracle as orc   # imported before the fork so that the workers share its pages
    orc.stft(synth(0, 0, 4096).astype(np.float64), orc.hamming_periodic(W), H)
    with mp.get_context("fork").Pool(n_workers) as pool:
        pool.map(_pool_worker, [(i, 0.0) for i in range(n_workers)], chunksize=1)   # start + warm every worker
        t0 = time.perf_counter()
        counts = pool.map(_pool_worker, [(i, seconds) for i in range(n_workers)], chunksize=1)
        wall = time.perf_counter() - t0
    total = int(sum(counts))
    print(json.dumps({"value": round(total * 441000 / wall / 1e6, 1), "unit": "Msamples/s", "cores": n_workers, "kind": "port",
                      "sample": f"{total} calls of the NumPy oracle stft on 10 s clips by a pool of {n_workers} processes, each looping for "
                                f"{seconds:.0f} s (wall {wall:.2f} s, {min(counts)}-{max(counts)} clips per worker), OMP/OPENBLAS threads = 1 per "
                                f"worker; os.cpu_count() = {os.cpu_count()}"}))


def cpu_baseline_all_cores(seconds=6.0):
    """Runs cpu_pool_main in a fresh interpreter (a fork pool must not inherit a HIP context)."""
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    try:
        res = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-pool", str(usable_cores()), str(seconds)], env=env,
                             capture_output=True, timeout=120, check=True)
        return json.loads(res.stdout.decode().strip().splitlines()[-1])
    except Exception as exc:   # the pool is a reported extra: never fail the bench line for it
        return {"error": f"{type(exc).__name__}: {exc}"}


def parity_probe(wl):
    """Normwise error of clip 0 and of the LAST clip of the batch vs the NumPy oracle (BASELINE metric: 'max |delta| vs
    NumPy'; SURVEY 8(d) tolerance: 1e-5 stft/istft/mdct/imdct, 1e-4 mel/mfcc/cqt)."""
    from oracle import zaf_oracle as orc
    kind, base, B = wl["kind"], wl["base"], wl["n_clips"]
    kind = kind[:-len("_compact")] if kind.endswith("_compact") else kind[:-len("_padded")] if kind.endswith("_padded") else kind
    ham, kbd = orc.hamming_periodic(W), orc.kbd_window(W)
    x64 = base[0].astype(np.float64)
    if kind in ("stft", "stft1", "stft_offgrid", "stftmag", "stftmag_offgrid", "stft64"):
        ref = orc.stft(x64, ham, H)
        ref = ref[:W // 2 + 1] if kind in ("stft1", "stftmag", "stftmag_offgrid") else ref
        ref = np.abs(ref) if kind.startswith("stftmag") else ref
    elif kind == "stft4096":
        ref = orc.stft(x64, orc.hamming_periodic(4096), 2048)
    elif kind == "stft4096_h1024":
        ref = orc.stft(x64, orc.hamming_periodic(4096), 1024)
    elif kind == "mdct4096":
        ref = orc.mdct(x64, orc.kbd_window(4096))
    elif kind == "stft8192":
        ref = orc.stft(x64, orc.hamming_periodic(8192), 4096)
    elif kind == "mdct8192":
        ref = orc.mdct(x64, orc.kbd_window(8192))
    elif kind == "mel4096":
        ref = orc.melspectrogram(x64, orc.hamming_periodic(4096), 2048, orc.melfilterbank(FS, 4096, 128))
    elif kind in ("istft", "istft1", "istft4096", "istft_offgrid", "istft64", "imdct64", "istft8192", "imdct8192"):
        ref = None   # (checked as a round trip below: the device spectrum is the input)
    elif kind in ("mdct", "mdct_offgrid", "mdct64", "mdct_pcm16"):
        ref = orc.mdct(x64, kbd)
    elif kind == "mel_pcm16":
        ref = orc.melspectrogram(x64, ham, H, orc.melfilterbank(FS, W, 128))
    elif kind in ("dct", "dct1000"):
        ref = orc.dct(x64, 2)
    elif kind in ("imdct", "imdct_offgrid"):
        ref = None
    elif kind == "mel_mfcc":   # both outputs, each against its own reference (the worse of the two is reported)
        fb = orc.melfilterbank(FS, W, 128)
        refs = (orc.melspectrogram(x64, ham, H, fb), orc.mfcc(x64, ham, H, fb, 20))
        first, last = wl["d_out"].download(0, 1)[0], wl["d_out"].download(B - 8, 1)[0]
        rels = [float(np.max(np.abs(g - r)) / np.max(np.abs(r))) for g, r in ((first[:128], refs[0]), (first[128:], refs[1]))]
        return {"replicas_bit_identical": bool(np.array_equal(first, last)), "max_rel_err_vs_numpy": max(rels), "max_rel_err_mel": rels[0], "max_rel_err_mfcc": rels[1],
                "tolerance": 1e-4, "within_tolerance": bool(max(rels) <= 1e-4)}
    elif kind in ("mel", "mfcc", "mel64", "mfcc64"):
        fb = orc.melfilterbank(FS, W, 128)
        ref = orc.melspectrogram(x64, ham, H, fb) if kind.startswith("mel") else orc.mfcc(x64, ham, H, fb, 20)
    elif kind in ("cqt", "cqt64"):
        ref = orc.cqtspectrogram(x64, FS, 25, orc.cqtkernel(FS, 24, 55, 3520))
    else:
        return {}
    first, last = wl["d_out"].download(0, 1)[0], wl["d_out"].download(B - 8, 1)[0]   # clip B-8 is a replica of clip 0
    out = {"replicas_bit_identical": bool(np.array_equal(first, last))}
    if ref is None:   # inverse kinds: resynthesis of the input (zaf.py:165-194 COLA; zaf.py:1098-1109 TDAC)
        n = len(x64) if kind.startswith("istft") else len(x64) - 1
        d = float(np.max(np.abs(first[:n].astype(np.float64) - x64[:n])))
        tol = 1e-11 if kind.endswith("64") else 1e-5
        out.update({"roundtrip_max_abs_residual": d, "tolerance": tol, "within_tolerance": bool(d < tol)})
        if kind.startswith("imdct"):
            kw = orc.kbd_window(8192) if kind == "imdct8192" else kbd
            refy = orc.imdct(orc.mdct(x64, kw), kw)
            e = float(np.max(np.abs(first - refy)) / np.max(np.abs(refy)))
            out.update({"max_rel_err_vs_numpy": e, "within_tolerance": bool(d < tol and e <= (1e-12 if kind.endswith("64") else 1e-5))})
        return out
    if first.ndim == 2 and first.shape[0] == ref.shape[0] and first.shape[1] > ref.shape[1]:
        first = first[:, :ref.shape[1]]   # (rows padded to 128-byte lines on the device)
    got = first if first.shape == ref.shape else first.T
    d = float(np.max(np.abs(got - ref)))
    tol = 1e-4 if kind in ("mel", "mfcc", "cqt", "mel4096", "mel_pcm16") else 1e-10 if kind == "mfcc64" else 1e-12 if kind.endswith("64") else 1e-5
    rel = d / float(np.max(np.abs(ref)))
    out.update({"max_abs_err_vs_numpy": d, "max_rel_err_vs_numpy": rel, "tolerance": tol, "within_tolerance": bool(rel <= tol)})
    return out


# ---------------------------------------------------------------------------------------------------------------
# timing
# ---------------------------------------------------------------------------------------------------------------
def prewarm(plan, d_in, d_out, n_clips, n_in, seconds=None):
    """Keep the device busy with this plan for `seconds` (untimed): the clocks of an idle MI355X ramp over the first few hundred
    milliseconds of work, and the contract's warm-up is a COUNT -- 5 launches of a 1 ms kernel end inside the ramp (round 2:
    the driver's --steps 20 --warmup 5 measured the ~1 ms kernels 5-8 % above their steady state)."""
    seconds = float(os.environ.get("ZAFX_BENCH_PREWARM_S", "0.0" if os.environ.get("ZAFX_BENCH_INNER") == "1" else "0.4")) if seconds is None else seconds
    t_end = time.perf_counter() + seconds
    n = 0
    while time.perf_counter() < t_end:
        for _ in range(4):
            plan.execute(d_in, d_out, n_clips, n_in)
        plan.sync()
        n += 4
    return n


def time_workload(wl, steps, warmup, rdzv):
    """Contract timing: W untimed steps, then EXACTLY K steps between (device sync + barrier) pairs, MAX over ranks;
    the HIP-event stopwatch of the plan's stream wraps the same K launches.  A second, separate pass times the K
    launches one by one (min / median of the kernel duration)."""
    plan, B, n_in = wl["plan"], wl["n_clips"], wl["n_in"]

    def sync_all():
        plan.sync()
        if rdzv is not None:
            rdzv.barrier()

    n_pre = prewarm(plan, wl["d_in"], wl["d_out"], B, n_in)   # time-based, untimed, in front of the contract's warm-up
    for _ in range(warmup):
        plan.execute(wl["d_in"], wl["d_out"], B, n_in)
    sync_all()
    t0 = time.perf_counter()
    plan.timer_start()
    for _ in range(steps):
        plan.execute(wl["d_in"], wl["d_out"], B, n_in)
    kernel_ms = plan.timer_stop() / max(steps, 1)
    sync_all()
    elapsed = time.perf_counter() - t0
    each = []
    for _ in range(steps):
        plan.timer_start()
        plan.execute(wl["d_in"], wl["d_out"], B, n_in)
        each.append(plan.timer_stop())
    vals = [elapsed, kernel_ms, -min(each), float(np.median(each))]
    per_rank = [kernel_ms]
    if rdzv is not None:
        import struct
        per_rank = [struct.unpack("<d", b)[0] for b in rdzv.all_gather(struct.pack("<d", kernel_ms))]
        vals = rdzv.all_reduce_max(vals)   # (-min: the MAX over ranks of -min is the smallest step anywhere)
    return {"elapsed_s": vals[0], "kernel_ms": vals[1], "kernel_ms_min": -vals[2], "kernel_ms_median": vals[3],
            "kernel_ms_per_rank": [round(v, 4) for v in per_rank],
            "segments": [["prewarm", n_pre], ["warmup", warmup], ["timed", steps], ["each", steps]]}


INNER_LOG = []   # {"kind", "kernel", "launches", "segments"} in launch order, written to $ZAFX_BENCH_INNER_LOG (profiled runs: which dispatches were the timed ones)


def live_traffic(kinds):
    """HBM bytes per launch of every kind's dominant kernel, measured NOW: two separate `rocprofv3 --pmc` passes (FETCH_SIZE,
    WRITE_SIZE: the TCC block cannot count both in one pass) of this script with --steps 3 over the same kinds, corrected as
    MI355X_MICROARCH.md prescribes (gfx950 tallies the 128-byte requests of streaming reads at 64 bytes: FETCH_SIZE x 2;
    WRITE_SIZE as counted -- both checked on the device-to-device copies of the same run, whose byte count is known).  The
    inner run logs its launches in order (kind, kernel, count); the counter rows are dealt to the kinds in that order, so two
    kinds that share a kernel (mel / mfcc) stay apart.  {kind: {...}}; {} when rocprofv3 is not there or fails."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None or os.environ.get("ZAFX_BENCH_LIVE_TRAFFIC", "1") == "0":
        return {}
    if any(k.startswith("ROCPROF") for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return {}   # this process is being profiled itself: no profiler inside a profiler
    out = {k: {} for k in kinds}
    work = tempfile.mkdtemp(prefix="zafx_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(work, counter)
            log = os.path.join(work, counter + ".log")
            env = dict(os.environ, TMPDIR="/tmp", ZAFX_BENCH_INNER="1", ZAFX_BENCH_INNER_LOG=log)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--",
                   sys.executable, os.path.abspath(__file__), "--kind", ",".join(kinds), "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300, check=True)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                return {}
            rows = []
            with open(files[0]) as fh:
                for r in csv.DictReader(fh):
                    if r.get("Counter_Name") == counter:
                        rows.append((int(r.get("Dispatch_Id", len(rows))), r["Kernel_Name"].split("(")[0], float(r["Counter_Value"])))
            rows.sort()
            with open(log) as fh:
                launches = json.load(fh)
            pos = 0
            for ent in launches:
                got = []
                while len(got) < ent["launches"] and pos < len(rows):
                    if ent["kernel"] in rows[pos][1]:
                        got.append(rows[pos][2])
                    pos += 1
                if ent["kind"] in out and got:
                    out[ent["kind"]][counter] = sum(got) / len(got) * 1024.0   # KB -> bytes, mean over the dispatches
                    out[ent["kind"]][counter + "_dispatches"] = len(got)
            copies = [v for _, k, v in rows if "copyBuffer" in k]
            if copies:   # make_workload replicates blocks of 8 clips device to device: dispatches of known size
                true = [441000 * 4 * 8, 1323000 * 4 * 8]
                ratios = sorted({round(min(true, key=lambda t: abs(t / (v * 1024.0) - 1.0)) / (v * 1024.0), 3) for v in copies if v > 0})
                out["_copy_check"] = dict(out.get("_copy_check", {}), **{counter + "_true_over_counter": ratios[:6]})
    except Exception as exc:   # reported, never fatal
        return {"_error": f"{type(exc).__name__}: {exc}"[:300]}
    finally:
        shutil.rmtree(work, ignore_errors=True)
    for k in kinds:
        if "FETCH_SIZE" in out[k] and "WRITE_SIZE" in out[k]:
            out[k]["hbm_bytes_per_launch"] = out[k]["FETCH_SIZE"] * 2.0 + out[k]["WRITE_SIZE"]
    return out


def apply_live_traffic(entry, live):
    """roofline.traffic of one entry from this run's counters."""
    roof = entry["roofline"]
    if live and "hbm_bytes_per_launch" in live:
        roof["traffic"] = round(live["hbm_bytes_per_launch"])
        roof["traffic_source"] = ("measured in this run: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py --steps 3` over the same "
                                  "kinds, started by this script; FETCH_SIZE x 2 + WRITE_SIZE (MI355X_MICROARCH.md, HBM section)")
        roof["traffic_detail"] = {k: (round(v) if isinstance(v, float) and v > 10 else v) for k, v in live.items()}
        roof["traffic_over_algorithmic"] = round(live["hbm_bytes_per_launch"] / roof["algorithmic_bytes_per_launch"], 4)


def roofline_of(wl, tm, kind):
    kernel_ms = tm["kernel_ms"]
    gbs = wl["bytes_per_launch"] / (kernel_ms * 1e-3) / 1e9
    roof = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
            "frac_of_achievable_6290": round(gbs / HBM_ACHIEVABLE_GBS, 4), "traffic": None,
            "kernel": wl["plan"].last_kernel, "kernel_ms": round(kernel_ms, 4), "kernel_ms_min": round(tm["kernel_ms_min"], 4),
            "kernel_ms_median": round(tm["kernel_ms_median"], 4), "algorithmic_bytes_per_launch": wl["bytes_per_launch"]}
    pmc = os.path.join(ROOT, "profiles", f"pmc_{kind}.json")
    if os.path.exists(pmc):   # a separate rocprofv3 --pmc collection of the same command, committed under profiles/
        with open(pmc) as f:
            rec = json.load(f)
        roof["traffic"] = rec.get("hbm_bytes_per_launch")
        roof["traffic_source"] = f"profiles/pmc_{kind}.json ({rec.get('collected', 'separate rocprofv3 --pmc passes')}); not measured in this run"
    sec = kernel_ms * 1e-3
    if kind == "cqt":
        # bound by the f32 VECTOR pipe (the contraction's matrix-core share is 13 instructions per wave
        # and frame).  `achieved` prices the flops 8(d) defines (the reference's complex transform); the real-input form
        # the kernel runs needs under half of them: both fractions, never summed with anything else.
        tf = wl["valu_flops"] / sec / 1e12
        roof.update({"bound": "valu", "achieved": round(tf, 2), "peak": F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / F32_PEAK_TFLOPS, 4),
                     "valu_frac_real_input": round(wl["valu_flops_real_input"] / sec / 1e12 / F32_PEAK_TFLOPS, 4),
                     "algorithmic_flops_per_launch": wl["valu_flops"], "real_input_flops_per_launch": wl["valu_flops_real_input"],
                     "flops": wl["flops_note"], "hbm": {"achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)}})
    elif kind == "cqt64":   # the float64 vector pipe (78.6 TF: MI355X_MICROARCH.md), flops as for cqt
        tf = wl["valu_flops"] / sec / 1e12
        roof.update({"bound": "valu", "achieved": round(tf, 2), "peak": F64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / F64_PEAK_TFLOPS, 4),
                     "valu_frac_real_input": round(wl["valu_flops_real_input"] / sec / 1e12 / F64_PEAK_TFLOPS, 4),
                     "hbm": {"achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)}})
    elif wl["mfma_flops"]:
        # with the filterbank's zero K-tiles skipped the binding roof is HBM (`frac`); how busy the two
        # arithmetic pipes are is reported beside it, each against its own 157.3 TF peak
        roof.update({"mfma_frac": round(wl["mfma_flops"] / sec / 1e12 / F32_PEAK_TFLOPS, 4),
                     "valu_frac": round(wl["valu_flops"] / sec / 1e12 / F32_PEAK_TFLOPS, 4),
                     "issued_mfma_flops_per_launch": wl["mfma_flops"], "valu_flops_per_launch": wl["valu_flops"], "flops": wl["flops_note"]})
    return roof


def e2e_pcie(device, clips=512):
    """PCIe-inclusive figures of the host-array boundary (never `value`): page-locked host f32 -> HBM -> kernel -> page-locked
    host c64 through Plan.run_host (zafx_run_host: chunks through a three-stage pipeline -- upload stream, kernel stream,
    download stream -- so that the three phases overlap), two-sided and one-sided, with the serial one-stream sequence of
    round 2 beside it."""
    import zafx
    N = 441000
    x = zafx.pinned_empty((clips, N), np.float32)
    x[:] = np.tile(np.stack([synth(0, c, N) for c in range(8)]), (clips // 8, 1))
    out = {}
    for label, onesided in (("two_sided", False), ("one_sided", True)):
        plan = zafx.stft_plan(zafx.hamming(W), H, device=device, onesided=onesided)
        host = zafx.pinned_empty(plan.out_shape(clips, N), plan.out_dtype)
        plan.run_host(x, N, out=host)   # (first call: staging buffers, page tables)
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            plan.run_host(x, N, out=host)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        rec = {"value": round(clips * N / best / 1e6, 1), "unit": "Msamples/s",
               "sample": f"{clips} clips x 10 s, page-locked host arrays both ways (zafx.pinned_empty), Plan.run_host = zafx_run_host: chunked, "
                         f"upload / kernel / download streams, best of 3: {best * 1e3:.1f} ms for {x.nbytes / 1e6:.0f} MB up + {host.nbytes / 1e6:.0f} MB down "
                         f"({host.nbytes / best / 1e9:.1f} GB/s of download alone)"}
        if not onesided:   # the serial sequence of round 2, same buffers
            d_in = zafx.DeviceBuffer((clips, N), np.float32, device)
            d_out = zafx.DeviceBuffer(plan.out_shape(clips, N), plan.out_dtype, device)
            serial = None
            for _ in range(2):
                t0 = time.perf_counter()
                d_in.upload(x)
                plan.execute(d_in, d_out, clips, N)
                plan.sync()
                d_out.download(out=host)
                dt = time.perf_counter() - t0
                serial = dt if serial is None else min(serial, dt)
            d_in.free()
            d_out.free()
            rec["serial_one_stream"] = {"value": round(clips * N / serial / 1e6, 1), "unit": "Msamples/s"}
        out[label] = rec
        del host
    # the transforms with small outputs are bound by the UPLOAD over PCIe -- melspectrogram from float32 host
    # arrays and from int16 PCM (Plan.run_host_pcm: the integers cross the link, zaf.py:1202 / :65 run on the device)
    plan = zafx.mel_plan(zafx.hamming(W), H, zafx.melfilterbank(FS, W, 128), device=device)
    host = zafx.pinned_empty(plan.out_shape(clips, N), plan.out_dtype)
    pcm = zafx.pinned_empty((clips, N), np.int16)
    pcm[:] = np.clip(np.round(x * 8192.0), -32768, 32767).astype(np.int16)
    for label, call in (("mel_f32", lambda: plan.run_host(x, N, out=host)), ("mel_pcm16", lambda: plan.run_host_pcm(pcm, out=host))):
        call()
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            call()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        out[label] = {"value": round(clips * N / best / 1e6, 1), "unit": "Msamples/s",
                      "sample": f"{clips} clips x 10 s, page-locked host arrays, best of 3: {best * 1e3:.1f} ms"}
    out["mel_pcm16_over_f32"] = round(out["mel_pcm16"]["value"] / out["mel_f32"]["value"], 3)
    out["value"] = out["two_sided"]["value"]
    out["unit"] = "Msamples/s"
    return out


def run_kind(kind, args, device, rank, world, rdzv, comm, with_cpu):
    """One config: build the device-resident workload, broadcast its constants, time it, probe parity -> (entry, timing, workload info)."""
    import zafx
    wl = make_workload(kind, device, args.layout)
    bcast = "none (1 rank)" if world == 1 else "none (no RCCL communicator: every rank built its own constants)"
    bcast_s = None
    if comm is not None:
        try:
            t0 = time.perf_counter()
            comm.broadcast_constants(wl["plan"], root=0)
            bcast_s = time.perf_counter() - t0
            bcast = "rccl ncclBroadcast of plan constants from rank 0"
        except zafx.ZafxError as exc:
            # every rank has already built identical constants from the same deterministic host code, so the measurement
            # stands; the failure is carried into the line (main() turns it into a non-zero exit)
            bcast = f"FAILED ({exc}); every rank built its own constants"
    tm = time_workload(wl, args.steps, args.warmup, rdzv)
    tm["broadcast_s"] = bcast_s
    inner = os.environ.get("ZAFX_BENCH_INNER") == "1"
    INNER_LOG.append({"kind": kind, "kernel": wl["plan"].last_kernel, "launches": sum(n for _, n in tm["segments"]), "segments": tm["segments"],
                      "kernel_ms": tm["kernel_ms"]})
    entry = None
    if rank == 0:
        total = float(wl["n_clips"]) * wl["samples_per_clip"] * world * args.steps
        entry = {"workload": wl["desc"], "value": round(total / tm["elapsed_s"] / 1e6, 1), "unit": "Msamples/s",
                 "ms_per_step": round(tm["elapsed_s"] / args.steps * 1e3, 4), "roofline": roofline_of(wl, tm, kind),
                 "parity": {} if inner else parity_probe(wl), "constants_broadcast": bcast}
        if world > 1 or comm is not None:
            entry["kernel_ms_per_rank"] = tm["kernel_ms_per_rank"]
            entry["constants_broadcast_wall_s"] = None if bcast_s is None else round(bcast_s, 4)
        if "placement" in wl:
            entry["placement"] = wl["placement"]
        if with_cpu:
            cb = cpu_baseline(kind, budget_s=10.0 if kind == "stft" else 3.0, min_calls=7 if kind == "stft" else 3)
            if cb:
                entry["cpu_baseline"] = cb
                entry["speedup_vs_cpu_baseline"] = round(entry["value"] / cb["value"], 1)
    info = {k: wl[k] for k in ("n_clips", "samples_per_clip", "desc")}
    free_workload(wl)
    return entry, tm, info


# ---------------------------------------------------------------------------------------------------------------
# the contract line: ONE compact JSON line, printed last, small enough that a record keeping only the tail of stdout
# still holds every BASELINE config (round 3's 18.8 KB line lost configs 3-5); everything else goes to bench_detail.json
# ---------------------------------------------------------------------------------------------------------------
LINE_LIMIT = 7000
_ROOF_KEEP = ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "kernel", "kernel_ms", "kernel_ms_median")


def _r(v, nd=4):
    return round(v, nd) if isinstance(v, float) else v


def compact_roofline(roof):
    out = {k: _r(roof[k]) for k in _ROOF_KEEP if k in roof}
    if isinstance(out.get("traffic"), float):
        out["traffic"] = round(out["traffic"])
    if "hbm" in roof:   # the kind priced against the f32 vector peak carries its HBM fraction beside it
        out["hbm_frac"] = roof["hbm"]["frac"]
    for k in ("mfma_frac", "valu_frac", "valu_frac_real_input"):
        if k in roof:
            out[k] = roof[k]
    return out


def compact_parity(par):
    out = {}
    for k in ("max_rel_err_vs_numpy", "roundtrip_max_abs_residual"):
        if k in par:
            out[{"max_rel_err_vs_numpy": "max_rel_err", "roundtrip_max_abs_residual": "roundtrip_max_abs"}[k]] = float(f"{par[k]:.3g}")
    for k in ("tolerance", "within_tolerance"):
        if k in par:
            out[k] = par[k]
    return out


def compact_entry(e):
    out = {"ms_per_step": e["ms_per_step"], "value": e["value"]}
    if "roofline" in e:
        out["roofline"] = compact_roofline(e["roofline"])
    if e.get("parity"):
        out["parity"] = compact_parity(e["parity"])
    if "residual_max_abs" in e:
        out["residual_max_abs"] = float(f"{e['residual_max_abs']:.3g}") if e["residual_max_abs"] is not None else None
    if "cpu_baseline" in e:
        out["cpu_baseline"] = {"value": e["cpu_baseline"]["value"], "cores": e["cpu_baseline"]["cores"]}
    return out


def compact_line(full):
    """The one line the driver reads, from the full record: contract keys, the headline's roofline / cpu_baseline /
    parity, one [ms, frac] pair per extra geometry, and -- LAST -- every BASELINE config with its own roofline, parity
    and cpu_baseline.  No prose beyond the names the contract asks for."""
    out = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "prewarm_launches", "ms_per_step", "higher_is_better", "scaling",
                                "vs_baseline", "dtype") if k in full}
    out["data"] = "synthetic white noise f32 (default_rng([0,c]).standard_normal); 8 distinct clips replicated on device to 1024 per GPU"
    cfg = full["config"]
    out["config"] = {k: cfg[k] for k in ("workload", "clips_per_gpu", "samples_per_clip", "parallelism", "layout", "allocator") if k in cfg}
    pl = cfg.get("placement")
    if isinstance(pl, dict) and "survey_probe_ms" in pl:
        out["config"]["placement_survey_ms"] = pl["survey_probe_ms"]
    out["roofline"] = compact_roofline(full["roofline"])
    if "cpu_baseline" in full:
        cb = full["cpu_baseline"]
        out["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"], "sample": cb["sample"][:160]}
        out["speedup_vs_cpu_baseline"] = full.get("speedup_vs_cpu_baseline")
    ac = full.get("cpu_baseline_all_cores")
    if ac and "value" in ac:
        out["cpu_baseline_all_cores"] = {"value": ac["value"], "cores": ac["cores"]}
    if full.get("parity"):
        out["parity"] = compact_parity(full["parity"])
    pc = full.get("end_to_end_pcie")
    if pc and "value" in pc:
        out["end_to_end_pcie"] = {k: (v["value"] if isinstance(v, dict) else v) for k, v in pc.items() if k not in ("unit",)}
    if "rccl" in full:
        out["rccl"] = full["rccl"]
    if "error" in full:
        out["error"] = full["error"][:300]
    out["detail"] = full.get("detail", "bench_detail.json")
    if full.get("extras"):   # [ms_per_step, roofline.frac, kernel]
        out["extras"] = {k: [e["ms_per_step"], e["roofline"]["frac"], e["roofline"]["kernel"]] for k, e in full["extras"].items()}
    if full.get("configs"):
        out["configs"] = {k: compact_entry(e) for k, e in full["configs"].items()}
    line = json.dumps(out, separators=(",", ":"))
    if len(line) > LINE_LIMIT:   # never outgrow the record: shed the optional keys in this order
        for k in ("extras", "end_to_end_pcie", "cpu_baseline_all_cores", "rccl"):
            if k in out and len(line) > LINE_LIMIT:
                del out[k]
                line = json.dumps(out, separators=(",", ":"))
    return line


def selftest_launch(launch):
    """The control flow of an N-rank run without a GPU (tests/test_launch.py): same rendezvous calls in the same order as
    main(), the oracle standing in for the device step on this rank's clip range."""
    import zafx
    from oracle import zaf_oracle as orc
    rank, local_rank, world = launch.rank_env()
    rdzv = launch.Rendezvous.from_env(timeout=120.0)
    uid = rdzv.broadcast(bytes(range(128)) if rank == 0 else b"")
    n_clips, n = 7, 3000
    lo, hi = zafx.clip_range(n_clips, rank, world)
    shard = np.stack([synth(4, c, n) for c in range(lo, hi)]).astype(np.float64) if hi > lo else np.zeros((0, n))
    rdzv.barrier()
    t0 = time.perf_counter()
    out = orc.stft_batch(shard, orc.hamming_periodic(256), 64)
    rdzv.barrier()
    elapsed = time.perf_counter() - t0
    vals = rdzv.all_reduce_max([elapsed, float(rank + 1)])
    sums = rdzv.all_gather(np.asarray([lo, hi, float(np.sum(np.abs(out)))], dtype=np.float64).tobytes())
    if rank == 0:
        parts = [np.frombuffer(b, dtype=np.float64).tolist() for b in sums]
        print(json.dumps({"n_gpus": world, "uid_ok": uid == bytes(range(128)), "max_rank_plus_1": vals[1], "elapsed_s": vals[0],
                          "device_of_rank": local_rank, "shards": parts}))
        sys.stdout.flush()
    rdzv.close()
    return 0


def main():
    if len(sys.argv) >= 4 and sys.argv[1] == "--cpu-pool":
        return cpu_pool_main(int(sys.argv[2]), float(sys.argv[3]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--kind", default="all", help="all = headline STFT + every other BASELINE config in one line; or one of "
                    "stft istft mdct imdct mel mfcc cqt dct or an extra: " + " ".join(EXTRA_KINDS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="with --kind all: headline only")
    ap.add_argument("--layout", default="FT", choices=["FT", "TF"], help="FT = reference (W, T) memory order (default); TF = frame-major")
    ap.add_argument("--selftest-launch", action="store_true", help="CPU only: run the N-rank control flow (rendezvous, id broadcast, "
                    "clip sharding with the oracle standing in for the device step, barrier, MAX) and print its JSON line")
    args = ap.parse_args()

    from zafx import launch
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: start the N ranks (one process per GPU) and relay rank 0's line
        code, out = launch.spawn_ranks([os.path.abspath(__file__)] + sys.argv[1:], args.gpus)
        sys.stdout.write(out)
        sys.stdout.flush()
        return code

    if args.selftest_launch:
        return selftest_launch(launch)

    # Libraries (RCCL banners, HIP warnings) write to C-level stdout; the contract is ONE JSON line
    # there, so keep the real stdout aside and send everything else to stderr.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import zafx
    rank, local_rank, world = launch.rank_env()
    force_dist = os.environ.get("ZAFX_BENCH_FORCE_DIST") == "1"   # exercise the N>1 plumbing on a 1-GPU box
    rdzv = launch.Rendezvous.from_env() if (world > 1 or force_dist) else None
    device = local_rank if world > 1 else 0
    if os.environ.get("ZAFX_BENCH_SHARE_DEVICES") == "1":   # test aid: N ranks on a box with fewer GPUs (RCCL then refuses the duplicate
        from zafx import _lib                                #           device and the constants stay per rank; everything else is the N-rank flow)
        device = local_rank % max(_lib.device_count(), 1)

    comm = None
    comm_hung = False
    rccl = None
    if rdzv is not None:
        # the path's only collective: RCCL broadcast of the shared constants from rank 0 over xGMI; the 128-byte id
        # of the communicator travels through the file rendezvous
        # (librccl prints a banner -- ROCm version, hostname, library path -- through C stdio on stdout; it would leave the C
        # buffer at exit, i.e. AFTER the JSON line.  File descriptor 1 points at stderr while the communicator is created and
        # the C buffers are flushed before it comes back: stdout carries the one JSON line and nothing else.)
        import ctypes
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        comm_error = None
        t_comm = time.perf_counter()
        try:
            uid = rdzv.broadcast(zafx.Comm.unique_id() if rank == 0 else b"")
            # ncclCommInitRank blocks until every rank has joined; created on a helper thread so that a bootstrap that never
            # completes (no usable network interface, a rank that died) ends in an error line, not in a hang
            import threading
            made = {}

            def create():
                try:
                    made["comm"] = zafx.Comm(device, rank, world, uid)
                except zafx.ZafxError as exc:
                    made["error"] = exc

            th = threading.Thread(target=create, daemon=True)
            th.start()
            th.join(timeout=float(os.environ.get("ZAFX_BENCH_COMM_TIMEOUT", "180")))
            if th.is_alive():
                comm_hung = True
                comm_error = "ncclCommInitRank did not return within the timeout"
            elif "error" in made:
                raise made["error"]
            else:
                comm = made["comm"]
        except zafx.ZafxError as exc:
            comm_error = str(exc)
        finally:
            try:
                ctypes.CDLL(None).fflush(None)
            except OSError:
                pass
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
        t_comm = time.perf_counter() - t_comm
        if comm_error:
            sys.stderr.write(f"rank {rank}: no RCCL communicator ({comm_error}); constants stay per-rank\n")
        # what RCCL itself says about the communicator (ncclCommCount / ncclCommUserRank), gathered from every rank
        seen, urank = (comm.count(), comm.user_rank()) if comm is not None else (0, -1)
        rows = [b.decode().split("|", 2) for b in rdzv.all_gather(f"{seen}|{urank}|{comm_error or ''}".encode())]
        ok = all(int(r[0]) == world for r in rows) and sorted(int(r[1]) for r in rows) == list(range(world))
        rccl = {"ok": ok, "ranks_seen": [int(r[0]) for r in rows], "user_ranks": [int(r[1]) for r in rows],
                "world_size": world, "comm_init_wall_s": round(t_comm, 3),
                "errors": sorted({r[2] for r in rows if r[2]})}
        # a collective needs every rank: if one of them has no communicator, nobody uses theirs
        if not ok:
            comm = None

    kinds = args.kind.split(",") if args.kind != "all" else ["stft"] + ([] if args.no_configs else list(CONFIG_KINDS) + list(EXTRA_KINDS))
    with_cpu = world == 1 and not args.no_cpu_baseline
    entries, extras = {}, {}
    head = head_info = head_tm = None
    for kind in kinds:
        entry, tm, info = run_kind(kind, args, device, rank, world, rdzv, comm, with_cpu and kind not in EXTRA_KINDS)
        if head is None:
            head, head_info, head_tm = entry, info, tm
        elif kind in EXTRA_KINDS:
            extras[kind] = entry
        else:
            entries[kind] = entry
    if comm is not None:
        comm.destroy()
    exit_code = 0
    if os.environ.get("ZAFX_BENCH_INNER_LOG"):
        with open(os.environ["ZAFX_BENCH_INNER_LOG"], "w") as fh:
            json.dump(INNER_LOG, fh)
    if rank == 0 and world == 1 and args.kind == "all" and os.environ.get("ZAFX_BENCH_INNER") != "1":
        # HBM traffic of the headline and of every BASELINE config from this run's own counters (two profiled child runs)
        measured = [k for k in kinds if k not in EXTRA_KINDS]
        live = live_traffic(measured)
        for k in measured:
            apply_live_traffic(head if k == kinds[0] else entries[k], live.get(k))
        if "_error" in live:
            head["roofline"]["traffic_live_error"] = live["_error"]
        if "_copy_check" in live:
            head["roofline"]["traffic_copy_check"] = live["_copy_check"]

    if rank == 0:
        hk = kinds[0]
        out = {
            "metric": "audio Msamples/sec (STFT win=2048 hop=1024)" if hk == "stft" else f"audio Msamples/sec ({hk})",
            "value": head["value"], "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "prewarm_launches": head_tm["segments"][0][1],   # untimed, time-based (ZAFX_BENCH_PREWARM_S), IN FRONT of the contract's warm-up
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64" if hk == "stft64" else "f32",
            "data": "synthetic white Gaussian noise (default_rng([0,c]).standard_normal, f32); 8 distinct clips replicated on device to 1024 per GPU",
            "config": {"workload": head["workload"], "clips_per_gpu": head_info["n_clips"], "samples_per_clip": head_info["samples_per_clip"],
                       "parallelism": f"clip-sharded x{world}", "constants_broadcast": head["constants_broadcast"],
                       "launcher": "file rendezvous (zafx/launch.py), no torch.distributed" if rdzv is not None else "single process",
                       "layout": "FT (reference memory order)" if args.layout == "FT" else "TF (frame-major)",
                       "placement": head.get("placement", "first allocation"),
                       "allocator": ("hipMalloc (ZAFX_ALLOC_CHUNK_MB=0)" if os.environ.get("ZAFX_ALLOC_CHUNK_MB", "64").strip() in ("0", "") else
                                     f"zafx_alloc: arrays of 1 GiB and more in {os.environ.get('ZAFX_ALLOC_CHUNK_MB', '64')}-MiB physical chunks (HIP virtual-memory API)")},
            "roofline": head["roofline"],
        }
        if rccl is not None:
            # the record of a multi-GPU run: what RCCL saw, not what the launcher said
            out["rccl"] = dict(rccl, kernel_ms_per_rank=head_tm["kernel_ms_per_rank"],
                               constants_broadcast_wall_s=None if head_tm.get("broadcast_s") is None else round(head_tm["broadcast_s"], 4))
            failed = [k for k, e in [(kinds[0], head)] + list(entries.items()) + list(extras.items()) if "FAILED" in e["constants_broadcast"]]
            if not rccl["ok"] or failed:
                out["error"] = ("RCCL communicator of %d ranks not formed (%s)" % (world, "; ".join(rccl["errors"]) or "rank counts differ")
                                if not rccl["ok"] else "RCCL broadcast failed for: " + ", ".join(failed))
                exit_code = 0 if os.environ.get("ZAFX_BENCH_ALLOW_NO_COMM") == "1" else 3
        out.update({k: v for k, v in head["parity"].items() if k.startswith("max_")})
        out["parity"] = head["parity"]
        if "cpu_baseline" in head:
            out["cpu_baseline"] = head["cpu_baseline"]
            out["speedup_vs_cpu_baseline"] = head["speedup_vs_cpu_baseline"]
        if entries:
            if "mdct" in entries and "imdct" in entries:   # BASELINE config 4: the pair, residual < 1e-5
                pair_ms = entries["mdct"]["ms_per_step"] + entries["imdct"]["ms_per_step"]
                entries["mdct_imdct_roundtrip"] = {
                    "ms_per_step": round(pair_ms, 4),
                    "value": round(head_info["n_clips"] * 441000.0 * world / (pair_ms * 1e-3) / 1e6, 1), "unit": "Msamples/s",
                    "residual_max_abs": entries["imdct"]["parity"].get("roundtrip_max_abs_residual"), "residual_bound": 1e-5,
                    "note": "imdct(mdct(x)) on the device, both kernels timed separately on the same 1024-clip batch; residual over clip 0 (zaf.py:1098-1109)"}
            out["configs"] = entries
        if extras:
            out["extras"] = extras
        if world == 1 and hk == "stft" and args.kind == "all" and "placement" not in head:
            try:
                out["config"]["placement"] = placement_survey(device, head["roofline"]["kernel_ms"], args.layout) or "first allocation"
            except zafx.ZafxError as exc:
                out["config"]["placement"] = {"timed_allocation": "first", "survey_error": str(exc)}
        if with_cpu and hk == "stft":
            out["cpu_baseline_all_cores"] = cpu_baseline_all_cores()
            if "value" in out["cpu_baseline_all_cores"]:
                out["speedup_vs_cpu_all_cores"] = round(out["value"] / out["cpu_baseline_all_cores"]["value"], 1)
            try:
                out["end_to_end_pcie"] = e2e_pcie(device)
            except zafx.ZafxError as exc:
                out["end_to_end_pcie"] = {"error": str(exc)}
        # the full record (workload prose, flop notes, placement survey, traffic sources) goes beside the script and to stderr;
        # stdout carries the ONE compact line, last
        detail = os.path.join(ROOT, "gpurun_out") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else ROOT
        detail = os.path.join(detail, "bench_detail.json")
        try:
            if os.environ.get("ZAFX_BENCH_INNER") == "1":
                raise OSError("profiled child run: no record")
            with open(detail, "w") as fh:
                json.dump(out, fh, indent=1)
            out["detail"] = os.path.relpath(detail, ROOT)
        except OSError:
            out["detail"] = "stderr"
        sys.stderr.write("bench detail: " + json.dumps(out) + "\n")
        sys.stderr.flush()
        sys.stdout.flush()
        os.write(real_stdout, (compact_line(out) + "\n").encode())

    if rdzv is not None:
        # every rank leaves with rank 0's verdict (a launcher reports the first non-zero exit)
        exit_code = int(rdzv.broadcast(str(exit_code).encode() if rank == 0 else b"").decode() or 0)
        rdzv.close()
    if comm_hung:   # a thread is still inside ncclCommInitRank: leave without the library's exit handlers
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(exit_code)
    return exit_code


if __name__ == "__main__":
    sys.exit(main())
