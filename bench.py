#!/usr/bin/env python3
"""Headline benchmark: batched STFT (BASELINE.json configs[1]) in audio Msamples/s.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--kind stft|istft|mdct|imdct|mel|mfcc|cqt]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch: 1024 clips x 10 s @ 44.1 kHz per GPU
(weak scaling: every rank transforms its own 1024 clips; clips are independent, there is
no data-path collective -- the only communication is the RCCL broadcast of the window /
filterbank constants from rank 0 before the timed region).  Inputs and outputs are
resident in HBM when the timed region starts.  Rank 0 prints ONE JSON line.

`roofline.achieved` = algorithmic bytes per launch / mean kernel duration measured with HIP
events on the plan's stream.  `cpu_baseline` = the NumPy oracle (a port of zaf.stft, same
NumPy calls per clip) timed on this host, rank 0, N=1 only, bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "zaf-python_amd"))
sys.path.insert(0, ROOT)

import zafx  # noqa: E402

FS, W, H = 44100, 2048, 1024
HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F32_PEAK_TFLOPS = 157.3    # dense f32 vector / f32-input MFMA peak


def synth(seed, c, n):
    return np.random.default_rng([seed, c]).standard_normal(n).astype(np.float32)


def make_workload(kind, device, layout="FT"):
    """Returns dict(plan, n_clips, n_in, d_in, d_out, samples_per_clip, bytes_per_launch, flops_per_launch, desc)."""
    ham = zafx.hamming(W)
    kbd = zafx.kaiser_bessel_derived(W)
    B, N = 1024, 441000
    distinct = 8
    T = 432
    if kind == "cqt":
        B, N, T = 1024, 1323000, 750   # BASELINE config 5: 8192 clips x 30 s over 8 GPUs = 1024 per GPU
    if kind == "dct":
        B, N, T = 16384, 1024, 1
    if kind == "stft64":
        B = 128
    base = np.stack([synth(0, c, N) for c in range(distinct)])
    d_base = zafx.DeviceBuffer.from_host(base, device)
    d_x = zafx.DeviceBuffer((B, N), np.float32, device)
    for r in range(B // distinct):
        d_x.copy_from(d_base, dst_offset=r * distinct * N * 4)
    d_base.free()
    wl = dict(n_clips=B, samples_per_clip=N, base=base, flops_per_launch=0.0)
    if kind == "stft":
        plan = zafx.stft_plan(ham, H, layout=layout, device=device)
        wl.update(plan=plan, d_in=d_x, n_in=N, bytes_per_launch=B * (4 * N + 8 * W * T),
                  desc="Batched STFT: 1024 clips x 10 s @ 44.1 kHz, Hamming win=2048 hop=1024, two-sided c64 (W,T) layout")
    elif kind == "stft64":  # float64 device arithmetic (written for exactness, not speed)
        d_x64 = zafx.DeviceBuffer.from_host(np.tile(base.astype(np.float64), (B // distinct, 1)), device)
        d_x.free()
        d_x = d_x64
        plan = zafx.stft_plan(ham, H, layout=layout, device=device, f64=True)
        wl.update(plan=plan, d_in=d_x, n_in=N, bytes_per_launch=B * (8 * N + 16 * W * T),
                  desc="Batched STFT in float64 / complex128: 128 clips x 10 s, Hamming win=2048 hop=1024, two-sided")
    elif kind == "stft1":   # one-sided output (rows 0..W/2), not the headline
        plan = zafx.stft_plan(ham, H, layout=layout, device=device, onesided=True)
        wl.update(plan=plan, d_in=d_x, n_in=N, bytes_per_launch=B * (4 * N + 8 * (W // 2 + 1) * T),
                  desc="Batched STFT, one-sided output (W/2+1, T): 1024 clips x 10 s, Hamming win=2048 hop=1024")
    elif kind == "istft1":
        fwd = zafx.stft_plan(ham, H, device=device, onesided=True)
        d_s = zafx.DeviceBuffer(fwd.out_shape(B, N), np.complex64, device)
        fwd.execute(d_x, d_s, B, N)
        fwd.sync()
        d_x.free()
        plan = zafx.istft_plan(ham, H, device=device, onesided=True)
        wl.update(plan=plan, d_in=d_s, n_in=T, bytes_per_launch=B * (8 * (W // 2 + 1) * T + 4 * (T * H - (W - H))),
                  desc="Batched ISTFT from one-sided spectra: 1024 clips x 432 frames, win=2048 hop=1024")
    elif kind == "istft":
        fwd = zafx.stft_plan(ham, H, device=device)
        d_s = zafx.DeviceBuffer(fwd.out_shape(B, N), np.complex64, device)
        fwd.execute(d_x, d_s, B, N)
        fwd.sync()
        d_x.free()
        plan = zafx.istft_plan(ham, H, device=device)
        wl.update(plan=plan, d_in=d_s, n_in=T, bytes_per_launch=B * (8 * W * T + 4 * (T * H - (W - H))),
                  desc="Batched ISTFT: 1024 clips x 432 frames, win=2048 hop=1024")
    elif kind == "mdct":
        plan = zafx.mdct_plan(kbd, device=device)
        wl.update(plan=plan, d_in=d_x, n_in=N, bytes_per_launch=B * (4 * N + 4 * (W // 2) * T),
                  desc="Batched MDCT: 1024 clips x 10 s, KBD win=2048")
    elif kind == "imdct":
        fwd = zafx.mdct_plan(kbd, device=device)
        d_m = zafx.DeviceBuffer(fwd.out_shape(B, N), np.float32, device)
        fwd.execute(d_x, d_m, B, N)
        fwd.sync()
        d_x.free()
        plan = zafx.mdct_plan(kbd, device=device, inverse=True)
        wl.update(plan=plan, d_in=d_m, n_in=T, bytes_per_launch=B * (4 * (W // 2) * T + 4 * ((W // 2) * (T - 1) - 1)),
                  desc="Batched IMDCT: 1024 clips x 432 frames, KBD win=2048")
    elif kind in ("mel", "mfcc"):
        fb = zafx.melfilterbank(FS, W, 128)
        rows = 128 if kind == "mel" else 20
        plan = zafx.mel_plan(ham, H, fb, None if kind == "mel" else 20, device=device)
        wl.update(plan=plan, d_in=d_x, n_in=N, bytes_per_launch=B * (4 * N + 4 * rows * T),
                  flops_per_launch=2.0 * 128 * 1024 * T * B,
                  desc=f"Fused {kind}: 1024 clips x 10 s, win=2048 hop=1024, 128 mel filters" + (", 20 coefficients" if kind == "mfcc" else ""))
    elif kind == "cqt":
        ck = zafx.cqtkernel(FS, 24, 55, 3520)
        plan = zafx.cqt_plan(FS, 25, ck, device=device)
        wl.update(plan=plan, d_in=d_x, n_in=N, bytes_per_launch=B * (4 * N + 4 * 144 * T),
                  flops_per_launch=B * T * 5.0 * 32768 * 15,
                  desc="cqtspectrogram: 1024 clips x 30 s @ 44.1 kHz per GPU (config 5: 8192 clips over 8 GPUs), 24 bins/octave 55-3520 Hz, 25 frames/s")
    elif kind == "dct":
        plan = zafx.linear_plan(zafx.dct_matrix(N, 2), device=device)
        wl.update(plan=plan, d_in=d_x, n_in=N, bytes_per_launch=B * 8 * N + 4 * N * N, flops_per_launch=2.0 * N * N * B,
                  desc="zaf.dct type 2 of 16384 vectors x 1024 samples as one f32 MFMA GEMM (SURVEY 8f rank 3)")
    else:
        raise SystemExit(f"unknown --kind {kind}")
    wl["d_out"] = zafx.DeviceBuffer(plan.out_shape(B, wl["n_in"]), plan.out_dtype, device)
    return wl


def cpu_baseline(budget_s=12.0):
    """zaf.stft restated with the same NumPy calls (oracle.stft), one 10 s clip per call, 1 core."""
    from oracle import zaf_oracle as orc   # checker / baseline only
    ham = orc.hamming_periodic(W)
    clips = [synth(0, c, 441000).astype(np.float64) for c in range(4)]
    for c in clips[:3]:
        orc.stft(c, ham, H)   # warm-up (first call pays FFT plan + page faults)
    times = []
    t_end = time.perf_counter() + budget_s
    i = 0
    while time.perf_counter() < t_end or len(times) < 7:
        t0 = time.perf_counter()
        orc.stft(clips[i % 4], ham, H)
        times.append(time.perf_counter() - t0)
        i += 1
    med = float(np.median(times))
    return {
        "value": round(441000 / med / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
        "sample": f"{len(times)} calls of the NumPy oracle stft (zaf.py:95-141 restated) on one 10 s clip each, "
                  f"median {med * 1e3:.2f} ms/clip, min {min(times) * 1e3:.2f} ms; numpy {np.__version__}; "
                  f"host has {os.cpu_count()} logical cores, FFT single-threaded",
    }


def parity_probe(wl, kind):
    """max |delta| of clip 0 vs the NumPy oracle (BASELINE metric: 'max |delta| vs NumPy')."""
    from oracle import zaf_oracle as orc
    if kind != "stft":
        return None
    got = wl["d_out"].download(0, 1)[0]
    if got.shape[0] != W:
        got = got.T
    ref = orc.stft(wl["base"][0].astype(np.float64), orc.hamming_periodic(W), H)
    d = float(np.max(np.abs(got - ref)))
    return {"max_abs_err_vs_numpy": d, "max_rel_err_vs_numpy": d / float(np.max(np.abs(ref)))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--kind", default="stft")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--layout", default="FT", choices=["FT", "TF"], help="FT = reference (W, T) memory order (default); TF = frame-major")
    args = ap.parse_args()

    # Libraries (RCCL banners, HIP warnings) write to C-level stdout; the contract is ONE JSON line
    # there, so keep the real stdout aside and send everything else to stderr.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    force_dist = os.environ.get("ZAFX_BENCH_FORCE_DIST") == "1"   # exercise the N>1 plumbing on a 1-GPU box
    if world > 1 or force_dist:
        import torch
        import torch.distributed as dist   # plumbing only: barrier + MAX over ranks
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    device = local_rank if world > 1 else 0

    wl = make_workload(args.kind, device, args.layout)
    plan = wl["plan"]

    bcast = "none (1 rank)"
    if world > 1 or force_dist:
        # the path's only collective: RCCL broadcast of the shared constants from rank 0 over xGMI
        # (every rank has already built identical constants from the same deterministic host code, so an error
        # here costs the demonstration of the collective, not the measurement)
        try:
            ids = [zafx.Comm.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            comm = zafx.Comm(device, rank, world, ids[0])
            comm.broadcast_constants(plan, root=0)
            comm.destroy()
            bcast = "rccl ncclBroadcast of plan constants from rank 0"
        except zafx.ZafxError as exc:
            bcast = f"skipped ({exc}); every rank built its own constants"

    def sync_all():
        plan.sync()
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()

    B, n_in = wl["n_clips"], wl["n_in"]
    for _ in range(args.warmup):
        plan.execute(wl["d_in"], wl["d_out"], B, n_in)
    sync_all()
    t0 = time.perf_counter()
    plan.timer_start()
    for _ in range(args.steps):
        plan.execute(wl["d_in"], wl["d_out"], B, n_in)
    kernel_ms = plan.timer_stop() / max(args.steps, 1)
    sync_all()
    elapsed = time.perf_counter() - t0

    if dist is not None:
        import torch
        t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(t[0]), float(t[1])

    if rank == 0:
        total_samples = float(B) * wl["samples_per_clip"] * world * args.steps
        value = total_samples / elapsed / 1e6
        achieved = wl["bytes_per_launch"] / (kernel_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", f"pmc_{args.kind}.json")
        if os.path.exists(pmc):
            with open(pmc) as f:
                traffic = json.load(f).get("hbm_bytes_per_launch")
        out = {
            "metric": "audio Msamples/sec (STFT win=2048 hop=1024)" if args.kind == "stft" else f"audio Msamples/sec ({args.kind})",
            "value": round(value, 1), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64" if args.kind == "stft64" else "f32",
            "data": "synthetic white Gaussian noise (default_rng([0,c]).standard_normal, f32); 8 distinct clips replicated on device to 1024 per GPU",
            "config": {"workload": wl["desc"], "clips_per_gpu": B, "samples_per_clip": wl["samples_per_clip"],
                       "parallelism": f"clip-sharded x{world}", "constants_broadcast": bcast, "layout": "FT (reference memory order)" if args.layout == "FT" else "TF (frame-major)"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": plan.kernel_name, "kernel_ms": round(kernel_ms, 4),
                         "algorithmic_bytes_per_launch": wl["bytes_per_launch"]},
        }
        if wl["flops_per_launch"]:
            tf = wl["flops_per_launch"] / (kernel_ms * 1e-3) / 1e12
            out["roofline"]["algorithmic_tflops"] = round(tf, 2)
            out["roofline"]["f32_peak_tflops"] = F32_PEAK_TFLOPS
            # configs 3 and 5 (mel / mfcc, cqt) and the GEMM are priced against the dense f32 peak (157.3 TF,
            # matrix and vector alike), not against HBM; the byte figures stay in the line as extra fields
            out["roofline"].update({"bound": "mfma", "achieved": round(tf, 2), "peak": F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": round(tf / F32_PEAK_TFLOPS, 4), "algorithmic_gbs": round(achieved, 1)})
        probe = parity_probe(wl, args.kind)
        if probe:
            out.update(probe)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            out["speedup_vs_cpu_baseline"] = round(value / out["cpu_baseline"]["value"], 1)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
