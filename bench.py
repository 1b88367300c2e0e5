#!/usr/bin/env python3
"""Headline benchmark: spectrogram frames/s of the U-Net forward on 513x256 fp32 batches (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one forward of the hot path over one resident batch per GPU (BASELINE configs[1]: batch 64 of
synthetic 513x256 fp32 spectrograms) + the per-clip CombinedPerceptualLoss kernels, and for N > 1 the one RCCL
all-gather of the per-clip values (4 floats per clip) (clips shard over ranks, weights replicated, no data-path collective; weak scaling: the batch
per GPU is fixed).  Rank 0 prints ONE JSON line; `value` = all ranks' frames / max-over-ranks time.

roofline: the dominant kernel is wino_conv_dma_f32 (the 17 3x3 convolutions, 95 % of the FLOPs; Winograd
F(2x2,3x3) on the fp32 matrix cores; ADN_CONV_ALGO=direct selects the direct implicit-GEMM kernel conv_mfma_f32).
Its launches are bracketed with hipEvents on the launch stream inside libadn (adn_unet_set_timing) during the timed
steps; achieved = ALGORITHMIC (direct-convolution) FLOPs of those launches / their summed duration (= average
FLOPs per launch / average launch duration), peak = 157.3 TFLOP/s (fp32 MFMA, MI355X_MICROARCH.md).  Winograd
executes 1/2.25 of the algorithmic FLOPs on the matrix cores, so `frac` may exceed 1; `mfma_util` is the share of
the matrix-core peak actually executed.  `traffic` (HBM bytes per launch from rocprofv3 PMC passes) is read from
profiles/pmc_traffic.json when that file has been produced.

cpu_baseline: the oracle's torch.nn.functional restatement of the reference forward (same ATen/oneDNN kernels
the reference's model.py dispatches to; kind "port") timed on this host's cores on a bounded sample.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

F_BINS, T_FRAMES = 513, 256


def host_cores() -> int:
    """CPU share of this process: min(affinity, cgroup quota); shared GPU boxes expose every core in the
    affinity mask but throttle by quota, and oversubscribing oneDNN threads collapses its throughput."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(sd_np, budget_s: float = 12.0):
    """Reference-equivalent CPU forward (oracle/unet_torch.py) on 513x256 clips, on this host's cores.

    The thread count is chosen by a short probe (one clip each at the candidate counts) so that the CPU is
    shown at its best; `cores` reports the threads actually used for the timed sample."""
    from oracle import unet_torch
    from audiodenoiser_amd.weights import make_input
    share = host_cores()
    sd = unet_torch.to_torch_state(sd_np)
    b = 2
    x = torch.from_numpy(make_input(0, b, F_BINS, T_FRAMES, scale=4.0))
    best_n, best_t = None, None
    for n in sorted({min(share, c) for c in (8, 16, 32, 64, share)}):
        torch.set_num_threads(n)
        unet_torch.unet_forward(sd, x[:1])      # warm-up at this thread count (oneDNN primitive creation)
        t0 = time.perf_counter()
        unet_torch.unet_forward(sd, x[:1])
        dt = time.perf_counter() - t0
        if best_t is None or dt < best_t:
            best_n, best_t = n, dt
        if dt > 6.0:
            break
    torch.set_num_threads(best_n)
    t0 = time.perf_counter()
    iters = 0
    while True:
        unet_torch.unet_forward(sd, x)
        iters += 1
        el = time.perf_counter() - t0
        if el >= budget_s or iters >= 16:
            break
    return {"value": round(iters * b * T_FRAMES / el, 1), "unit": "frames/s", "cores": best_n, "kind": "port",
            "sample": f"{iters} forwards of batch {b} x 513x256 fp32 ({el:.1f} s) with {best_n} threads "
                      f"(host share {share}), oracle/unet_torch.py = torch {torch.__version__} CPU/oneDNN, "
                      "the kernels the reference's model.py dispatches to"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch-per-gpu", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", choices=("f32", "f16"), default="f32",
                    help="f32 = headline metric (default); f16 = BASELINE configs[4] (fp16 storage + fp16 MFMA)")
    args = ap.parse_args()

    from audiodenoiser_amd import _lib
    from audiodenoiser_amd import distributed as D
    from audiodenoiser_amd.loss import perceptual_loss_per_clip
    from audiodenoiser_amd.model import UNet
    from audiodenoiser_amd.roofline import PEAK_HBM_GBS, PEAK_MFMA_F32_TFLOPS, unet_launches
    from audiodenoiser_amd.weights import make_state_dict

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device (no CPU path)")
    rank, local_rank, world = D.init_from_env("nccl")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)

    sd_np = make_state_dict(1234)                                     # replicated weights, regenerated per rank
    net = UNet(1, 1)
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd_np.items()}, strict=True)
    net = net.to(dev).eval().set_compute_dtype(args.dtype)
    f16 = args.dtype == "f16"

    b = args.batch_per_gpu
    g = torch.Generator(device=dev).manual_seed(rank)                 # per-rank shard of the global batch
    x = torch.rand((b, 1, F_BINS, T_FRAMES), generator=g, device=dev) * 4.0      # resident in HBM
    target = torch.rand((b, 1, F_BINS, T_FRAMES), generator=g, device=dev) * 4.0

    def step():
        y = net(x)
        loss = perceptual_loss_per_clip(y, target)            # (b, 4): total, stft, mel, l1 per clip
        return D.gather_per_clip(loss.reshape(-1))

    with torch.no_grad():
        for _ in range(args.warmup):
            allv = step()
        L = _lib.load()
        _lib.check(L.adn_unet_set_timing(net._handle, args.steps), "adn_unet_set_timing")
        D.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            allv = step()
        torch.cuda.synchronize(dev)
        D.barrier()
        elapsed = time.perf_counter() - t0
    elapsed = D.max_over_ranks(elapsed, dev)
    assert allv.numel() == 4 * b * world and bool(torch.isfinite(allv).all())

    # per-launch durations of the timed steps (events recorded on the launch stream inside libadn)
    ms = np.zeros((args.steps, 23), dtype=np.float32)
    for i in range(args.steps):
        _lib.check(L.adn_unet_get_timing(net._handle, i, ms[i].ctypes.data_as(_lib.c_float_p)), "adn_unet_get_timing")
    _lib.check(L.adn_unet_set_timing(net._handle, 0), "adn_unet_set_timing")
    ms_mean = ms.mean(axis=0)

    if rank == 0:
        launches = unet_launches(F_BINS, T_FRAMES)
        peak = 2516.6 if f16 else PEAK_MFMA_F32_TFLOPS      # dense MFMA peak of the arithmetic type (MI355X_MICROARCH.md)
        direct = f16 or os.environ.get("ADN_CONV_ALGO") == "direct"
        dom = [i for i, l in enumerate(launches) if l["kind"] == "conv3x3"]
        dom_flops = sum(launches[i]["flops"] for i in dom) * b          # per forward of this rank
        dom_ms = float(ms_mean[dom].sum())
        achieved = dom_flops / (dom_ms * 1e-3) / 1e12
        tot_flops = sum(l["flops"] for l in launches) * b
        tot_bytes = (sum(l["act_bytes"] for l in launches) * b + sum(l["weight_bytes"] for l in launches)) * (0.5 if f16 else 1.0)
        fwd_ms = float(ms_mean.sum())
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                with open(tpath) as fh:
                    traffic = json.load(fh).get("dominant_bytes_per_launch")
            except (OSError, ValueError):
                traffic = None
        frames = b * world * T_FRAMES * args.steps
        out = {
            "metric": "spectrogram frames/sec (forward), 513x256 fp32",
            "value": round(frames / elapsed, 1),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"batch={b} per GPU synthetic 513x256 fp32 spectrograms, full U-Net forward "
                                   + ("[fp16 storage + fp16 MFMA inside, BASELINE configs[4]] " if f16 else "")
                                   +
                                   "(BASELINE configs[1]) + per-clip perceptual loss" + (" + all-gather" if world > 1 else ""),
                       "batch_per_gpu": b, "global_batch": b * world, "freq_bins": F_BINS, "frames": T_FRAMES,
                       "parallelism": f"clips sharded over {world} rank(s), weights replicated"},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak,
                         "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": None if direct else traffic,
                         "kernel": (("conv_mfma<f16> (direct implicit GEMM, fp16 MFMA)" if f16 else
                                     "conv_mfma<f32> (direct implicit GEMM)") if direct
                                    else "wino_conv_dma_f32 (Winograd F(2x2,3x3), fp32 MFMA)") + ", 17 launches per forward",
                         "mfma_util": round(achieved / peak / (1.0 if direct else 2.25), 4),
                         "flops_per_launch": round(dom_flops / len(dom), 1),
                         "avg_launch_ms": round(dom_ms / len(dom), 4)},
            "forward": {"kernel_ms": round(fwd_ms, 3),
                        "tflops": round(tot_flops / (fwd_ms * 1e-3) / 1e12, 2),
                        "frac_mfma_peak": round(tot_flops / (fwd_ms * 1e-3) / 1e12 / peak, 4),
                        "algorithmic_GBps": round(tot_bytes / (fwd_ms * 1e-3) / 1e9, 1),
                        "frac_hbm_peak": round(tot_bytes / (fwd_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                        "per_launch_ms": {l["name"]: round(float(m), 4) for l, m in zip(launches, ms_mean)}},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(sd_np)
        print(json.dumps(out), flush=True)
    D.barrier()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
