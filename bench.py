#!/usr/bin/env python3
"""Headline benchmark: spectrogram frames/s of the U-Net forward on 513x256 fp32 batches (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one forward of the hot path over one resident batch per GPU (BASELINE configs[1]: batch 64 of synthetic
513x256 fp32 spectrograms) + the per-clip CombinedPerceptualLoss kernels, and for N > 1 the one RCCL all-gather of
the per-clip values (4 floats per clip).  Clips shard over ranks, weights are replicated, there is no data-path
collective; weak scaling: the batch per GPU is fixed.  Rank 0 prints ONE JSON line; `value` = all ranks' frames /
max-over-ranks time between barrier + synchronize pairs.

roofline (dominant kernel = wino4_conv_f32, Winograd F(4x4,3x3) on the exact-fp32 matrix cores: all 17 3x3 convolutions
at 513x256, 87 % of the forward's time; layers whose image its 32x32 tiles do not fit -- none at this size -- run
wino_conv_dma_f32, F(2x2,3x3), and would be summarised under `other_3x3_kernel`; ADN_WINO_TILE=2 pins F(2x2,3x3)
everywhere, ADN_CONV_ALGO=direct selects the direct implicit-GEMM kernel conv_mfma<float>):
  every launch of the timed steps is bracketed with hipEvents on the launch stream inside libadn
  (adn_unet_set_timing).  `achieved` = matrix-core FLOPs the kernel EXECUTES per launch (padded tiles counted;
  F(4x4,3x3) needs 36 multiply-adds per 4x4 output tile and channel pair, audiodenoiser_amd/roofline.py) / average
  launch duration; `peak` = 157.3 TFLOP/s dense fp32 MFMA (MI355X_MICROARCH.md); `frac` = achieved / peak <= 1.
  The ALGORITHMIC (direct-convolution) rate of the same launches is under `algorithmic`; it
  exceeds the direct-convolution roof because F(4x4,3x3) executes a quarter of those FLOPs.
  `traffic` = HBM bytes per launch from two separate rocprofv3 --pmc passes (2 x FETCH_SIZE + WRITE_SIZE, gfx950
  correction), read from profiles/pmc_traffic.json ONLY if that file was produced from this very build of libadn.so
  (source digest recorded in it); null otherwise.

Batch per GPU: 64 at N = 1 (BASELINE configs[1], the config `metric` is quoted on); 256 when launched on more than one
rank (BASELINE configs[3]: batch 2048 sharded over 8 GPUs = 256 clips per GPU; weak scaling, so 2 and 4 ranks also run 256
per GPU).  The N = 1 line carries `fp32_b256` -- the same step at 256 clips on one GPU -- which is the per-GPU figure the
N > 1 values are to be compared with.  Every line carries `ranks`: how many ranks the gathered tensor held, each rank's own
time for the K steps (min / max) and the all-gather's own time.

Sub-benchmarks in the same JSON line (rank 0, N = 1 only; --no-extras skips them):
  `fp32_b256`  the main step at batch 256 (north_star's batch, one rank's shard of configs[3]).
  `exact_f32`  the headline step with exact-fp32 matrix instructions everywhere (ADN_CONVT_SPLIT=0).
  `b1`    single-clip latency at 513x256 and the reference's own shapes (257x188 whole test set of 5 clips as test.py:112-113
          runs it, 256x64 at the training batch of 16): default (automatic small-grid kernels), ADN_BATCH_INVARIANT=1, and the
          explicit serving switches.
  `e2e_config0`  BASELINE configs[0] as one stream sequence: resident audio -> adn_stft_mag_fit -> adn_unet_forward at batch 1
          and 64, with the oracle chain's CPU time beside it.
  `host_boundary`  the PCIe-inclusive rates of the headline workload (spectrograms in host memory, as the reference's test.py
          hands them over): pageable `model(x_cpu)`, pinned on one stream, pinned with the copies on their own streams.
  `stft`  BASELINE configs[2]: 10 000 clips x 132 300 samples, n_fft 1024, hop 256, centred; HBM roofline of
          stft_wave_kernel (algorithmic bytes = audio read once + magnitudes written once = 1 590 084 B per clip),
          with the C oracle's STFT timed beside it.
  `f16`   BASELINE configs[4]: the same forward at batch 256 with fp16 storage + fp16 MFMA; MFMA roofline of
          conv16_f16 against the dense fp16 peak.

cpu_baseline: the oracle's torch.nn.functional restatement of the reference forward (same ATen/oneDNN kernels
the reference's model.py dispatches to; kind "port") timed on this host's cores on a bounded sample.
"""
from __future__ import annotations

import argparse
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


def lib_digest() -> str:
    """Code digest of the library this process runs.  A variant picked with ADN_LIBADN_PATH (tools/variant_bench.sh) is not the
    in-tree build: it reports "variant:<file>" so that no profile of the production build is attached to it."""
    variant = os.environ.get("ADN_LIBADN_PATH")
    if variant:
        return "variant:" + os.path.basename(variant)
    from audiodenoiser_amd import build as B
    return B.code_digest_of_built_library()


def tracked_traffic(kernel_key: str):
    """HBM bytes per launch of `kernel_key` from profiles/pmc_traffic.json if (and only if) that profile was taken
    with the library that is being benchmarked now."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as fh:
            rec = json.load(fh)
    except (OSError, ValueError):
        return None, "profiles/pmc_traffic.json missing"
    if lib_digest().startswith("variant:"):
        return None, "library variant (ADN_LIBADN_PATH): no PMC profile belongs to it"
    if rec.get("lib_digest") != lib_digest():
        return None, "profiles/pmc_traffic.json was taken with another build of libadn.so (digest differs)"
    ent = rec.get("kernels", {}).get(kernel_key)
    if not ent:
        return None, f"no PMC entry for {kernel_key}"
    return ent["bytes_per_launch"], (f"profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                     f"`{ent.get('cmd', '?')}` with this build (digest {rec['lib_digest'][:12]})")


def make_net(sd_np, dev, dtype):
    from audiodenoiser_amd.model import UNet
    net = UNet(1, 1)
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd_np.items()}, strict=True)
    return net.to(dev).eval().set_compute_dtype(dtype)


def launch_timings(net, steps):
    from audiodenoiser_amd import _lib
    L = _lib.load()
    ms = np.zeros((steps, 23), dtype=np.float32)
    for i in range(steps):
        _lib.check(L.adn_unet_get_timing(net._handle, i, ms[i].ctypes.data_as(_lib.c_float_p)), "adn_unet_get_timing")
    _lib.check(L.adn_unet_set_timing(net._handle, 0), "adn_unet_set_timing")
    return ms.mean(axis=0)


def conv_roofline(ms_mean, b, algo, peak, kernel_name, traffic_key, wino_mode="auto"):
    """MFMA roofline of the dominant 3x3 kernel from the event-timed durations of its launches (module docstring).

    fp32 Winograd path: the dominant kernel is wino4_conv_f32 (F(4x4,3x3)); 3x3 launches that stay on wino_conv_dma_f32
    (F(2x2,3x3): images its 32x32 tiles do not fit; none at 513x256) are summarised under `other_3x3_kernel`."""
    from audiodenoiser_amd.roofline import executed_mfma_flops, unet_launches, winograd_tile
    launches = unet_launches(F_BINS, T_FRAMES)
    conv = [i for i, l in enumerate(launches) if l["kind"] == "conv3x3"]
    dom = [i for i in conv if algo != "winograd" or winograd_tile(launches[i], wino_mode) == 4] or conv
    rest = [i for i in conv if i not in dom]

    def rates(idx):
        ms = float(ms_mean[idx].sum())
        executed = sum(executed_mfma_flops(launches[i], algo, wino_mode) for i in idx) * b
        algorithmic = sum(launches[i]["flops"] for i in idx) * b
        return ms, executed, algorithmic

    dom_ms, executed, algorithmic = rates(dom)
    ach = executed / (dom_ms * 1e-3) / 1e12
    alg = algorithmic / (dom_ms * 1e-3) / 1e12
    traffic, source = tracked_traffic(traffic_key)
    out = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
           "traffic": traffic, "traffic_source": source,
           "kernel": f"{kernel_name}, {len(dom)} launches per forward",
           "achieved_basis": "matrix-core FLOPs executed per launch (padded tiles counted) / event-timed launch duration",
           "executed_flops_per_launch": round(executed / len(dom), 1),
           "avg_launch_ms": round(dom_ms / len(dom), 4),
           "algorithmic": {"flops_per_launch": round(algorithmic / len(dom), 1), "tflops": round(alg, 2),
                           "frac_of_direct_conv_roof": round(alg / peak, 4),
                           "note": "direct-convolution FLOP count of SURVEY.md 8d; Winograd F(4x4,3x3) executes 1/4 of "
                                   "it, F(2x2,3x3) 1/2.25"},
           "algorithmic_bytes_per_launch": round(sum(launches[i]["act_bytes"] * b + launches[i]["weight_bytes"]
                                                     for i in dom) / len(dom) * (0.5 if algo == "direct_f16" else 1.0))}
    if rest:
        r_ms, r_exec, r_alg = rates(rest)
        out["other_3x3_kernel"] = {
            "kernel": "wino_conv_dma_f32 (Winograd F(2x2,3x3)): " + ", ".join(launches[i]["name"] for i in rest),
            "launches": len(rest), "avg_launch_ms": round(r_ms / len(rest), 4),
            "executed_mfma_tflops": round(r_exec / (r_ms * 1e-3) / 1e12, 2),
            "frac": round(r_exec / (r_ms * 1e-3) / 1e12 / peak, 4),
            "algorithmic_tflops": round(r_alg / (r_ms * 1e-3) / 1e12, 2)}
    return out


def convt_uses_split_bf16(f16: bool) -> bool:
    """fp32 transposed convolutions run on the bf16 matrix cores through a three-term split of both operands (six products, fp32
    accumulation; conv_dma<..., SPLIT>) unless ADN_CONVT_SPLIT=0 was set when the handle was created."""
    return (not f16) and os.environ.get("ADN_CONVT_SPLIT", "1") != "0"


def forward_summary(ms_mean, b, algo, peak, wino_mode="auto"):
    """All 23 launches.  `executed_mfma_tflops` / `frac_mfma_peak_executed` cover the launches that run on the matrix pipe `peak`
    belongs to (fp32 path: the 3x3 layers, and the transposed convolutions only when they run the exact-fp32 form); the
    split-bf16 transposed convolutions are summarised under `convt` against the bf16 peak."""
    from audiodenoiser_amd.roofline import ACHIEVABLE_HBM_GBS, PEAK_HBM_GBS, PEAK_MFMA_F16_TFLOPS, executed_mfma_flops, unet_launches
    launches = unet_launches(F_BINS, T_FRAMES)
    f16 = algo == "direct_f16"
    half = 0.5 if f16 else 1.0
    split = convt_uses_split_bf16(f16)
    fwd_ms = float(ms_mean.sum())
    tot_flops = sum(l["flops"] for l in launches) * b
    on_pipe = [i for i, l in enumerate(launches) if l["kind"] == "conv3x3" or (l["kind"] == "convt" and not split)]
    pipe_exec = sum(executed_mfma_flops(launches[i], algo if launches[i]["kind"] == "conv3x3" else ("direct_f16" if f16 else "direct"),
                                        wino_mode) for i in on_pipe) * b
    pipe_ms = float(ms_mean[on_pipe].sum())
    tot_bytes = (sum(l["act_bytes"] for l in launches) * b + sum(l["weight_bytes"] for l in launches)) * half
    out = {"kernel_ms": round(fwd_ms, 3),
           "algorithmic_tflops": round(tot_flops / (fwd_ms * 1e-3) / 1e12, 2),
           "executed_mfma_tflops": round(pipe_exec / (pipe_ms * 1e-3) / 1e12, 2),
           "frac_mfma_peak_executed": round(pipe_exec / (pipe_ms * 1e-3) / 1e12 / peak, 4),
           "executed_mfma_basis": f"{len(on_pipe)} launches on this matrix pipe, {round(pipe_ms, 3)} ms of the forward",
           "algorithmic_GBps": round(tot_bytes / (fwd_ms * 1e-3) / 1e9, 1),
           "frac_hbm_peak": round(tot_bytes / (fwd_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
           "frac_hbm_achievable": round(tot_bytes / (fwd_ms * 1e-3) / 1e9 / ACHIEVABLE_HBM_GBS, 4),
           "per_launch_ms": {l["name"]: round(float(m), 4) for l, m in zip(launches, ms_mean)}}
    # per launch: fraction of the matrix pipe's peak the launch EXECUTES at (padded tiles counted; the split-bf16 transposed
    # convolutions: six bf16 products against the bf16 peak) and fraction of the HBM roof its algorithmic bytes amount to
    # (8 TB/s spec / 6.3 TB/s achievable copy rate); a launch fused into its neighbour (~0 ms) is left out
    plf = {}
    for i, (l, m) in enumerate(zip(launches, ms_mean)):
        m = float(m)
        if m < 0.02:
            continue
        by = (l["act_bytes"] * b + l["weight_bytes"]) * half
        ent = {"hbm": round(by / (m * 1e-3) / 1e9 / PEAK_HBM_GBS, 3), "hbm_achievable": round(by / (m * 1e-3) / 1e9 / ACHIEVABLE_HBM_GBS, 3)}
        if l["kind"] == "conv3x3":
            ent["mfma"] = round(executed_mfma_flops(l, algo, wino_mode) * b / (m * 1e-3) / 1e12 / peak, 3)
        elif l["kind"] == "convt":
            ex = executed_mfma_flops(l, "direct_f16" if f16 else "direct") * b
            ent["mfma"] = round(ex * 6.0 / (m * 1e-3) / 1e12 / PEAK_MFMA_F16_TFLOPS, 3) if split else round(ex / (m * 1e-3) / 1e12 / peak, 3)
        plf[l["name"]] = ent
    out["per_launch_frac"] = plf
    ct = [i for i, l in enumerate(launches) if l["kind"] == "convt"]
    ct_ms = float(ms_mean[ct].sum())
    ct_alg = sum(launches[i]["flops"] for i in ct) * b
    ct_bytes = sum(launches[i]["act_bytes"] * b + launches[i]["weight_bytes"] for i in ct) * half
    out["convt"] = {"ms": round(ct_ms, 3), "algorithmic_tflops": round(ct_alg / (ct_ms * 1e-3) / 1e12, 2),
                    "algorithmic_GBps": round(ct_bytes / (ct_ms * 1e-3) / 1e9, 1),
                    "frac_hbm_peak": round(ct_bytes / (ct_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                    "frac_hbm_achievable": round(ct_bytes / (ct_ms * 1e-3) / 1e9 / ACHIEVABLE_HBM_GBS, 4)}
    if f16:
        out["convt"]["kernel"] = ("convt16_f16: v_mfma_f32_16x16x32_f16, items of 256 pixels x 256 columns, ring of four LDS images "
                                  "(copies three steps ahead), 16-byte stores from the accumulators (v_permlane16_swap)"
                                  if os.environ.get("ADN_F16_CONVT", "") != "dma" else "conv_dma<_Float16, 8, 128, ..., CONVT2X2>")
    if split:
        ct_exec = sum(executed_mfma_flops(launches[i], "direct") for i in ct) * b * 6.0
        out["convt"].update({
            "kernel": "conv_dma<float, 8, 128, 2, 2, 1, 2, CONVT, 3, SPLIT>: fp32 operands split into three bf16 terms, six "
                      "v_mfma_f32_32x32x16_bf16 products per term pair, fp32 accumulation (fp32-level accuracy)",
            "executed_bf16_mfma_tflops": round(ct_exec / (ct_ms * 1e-3) / 1e12, 2),
            "frac_bf16_mfma_peak": round(ct_exec / (ct_ms * 1e-3) / 1e12 / PEAK_MFMA_F16_TFLOPS, 4)})
    return out


def bench_stft(dev, clips=10000, length=132300, n_fft=1024, hop=256, steps=20, warmup=3, cpu_clips=4096):
    """BASELINE configs[2] on one GPU; inputs resident in HBM; events on the launch stream (torch's current stream)."""
    from audiodenoiser_amd import _lib
    from audiodenoiser_amd.stft import stft_n_frames
    g = torch.Generator(device=dev).manual_seed(0)
    a = torch.rand((clips, length), generator=g, device=dev) * 2 - 1
    nfr = stft_n_frames(length, n_fft, hop, True)
    nb = n_fft // 2 + 1
    out = torch.empty((clips, nb, nfr), dtype=torch.float32, device=dev)
    L = _lib.load()
    st = torch.cuda.current_stream(dev).cuda_stream

    def run():
        _lib.check(L.adn_stft_mag(a.data_ptr(), clips, length, n_fft, hop, 1, out.data_ptr(), st), "adn_stft_mag")
    for _ in range(warmup):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    e0.record()
    for _ in range(steps):
        run()
    e1.record()
    torch.cuda.synchronize(dev)
    ms = e0.elapsed_time(e1) / steps
    bytes_per_clip = length * 4 + nb * nfr * 4
    gbs = clips * bytes_per_clip / (ms * 1e-3) / 1e9
    traffic, source = tracked_traffic("stft_wave_kernel")
    res = {"workload": f"BASELINE configs[2]: {clips} clips x {length} samples (3 s @ 44.1 kHz), n_fft {n_fft}, hop {hop}, "
                       f"centred -> {clips} x {nb} x {nfr} fp32 magnitudes",
           "steps": steps, "warmup": warmup, "ms_per_launch": round(ms, 4),
           "clips_per_s": round(clips / (ms * 1e-3), 1), "frames_per_s": round(clips * nfr / (ms * 1e-3), 1),
           "dtype": "f32",
           "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s",
                        "frac": round(gbs / 8000.0, 4), "frac_of_achievable": round(gbs / 6300.0, 4),
                        "achievable": "6.3 TB/s: the measured float4 copy rate of MI355X_MICROARCH.md (79 % of the spec)",
                        "traffic": traffic, "traffic_source": source,
                        "kernel": "stft_wave_kernel<512,4,16,3>, 1 launch per step",
                        "algorithmic_bytes_per_launch": clips * bytes_per_clip, "bytes_per_clip": bytes_per_clip}}
    # the same clips straight to the network's input format (wav -> forward flow, BASELINE configs[0] at scale): STFT +
    # fp16 round trip + crop to 513x256 in one kernel (adn_stft_mag_fit) -- only the 256 frames the network reads
    fit = torch.empty((clips, 1, F_BINS, T_FRAMES), dtype=torch.float32, device=dev)

    def run_fit():
        _lib.check(L.adn_stft_mag_fit(a.data_ptr(), clips, length, n_fft, hop, 1, fit.data_ptr(), F_BINS, T_FRAMES, st),
                   "adn_stft_mag_fit")
    for _ in range(warmup):
        run_fit()
    torch.cuda.synchronize(dev)
    e0.record()
    for _ in range(steps):
        run_fit()
    e1.record()
    torch.cuda.synchronize(dev)
    ms_fit = e0.elapsed_time(e1) / steps
    fit_bytes = clips * ((T_FRAMES - 1) * hop + n_fft // 2 + F_BINS * T_FRAMES) * 4       # samples the 256 frames touch + output
    res["fused_to_network_input"] = {
        "what": f"adn_stft_mag_fit: the same clips -> ({clips},1,{F_BINS},{T_FRAMES}) = STFT + fp16 round trip + crop "
                "(data_loader.py:41-42,54-72) in one kernel, bit-identical to adn_stft_mag + adn_quantize_pad",
        "kernel": "stft_fit_kernel<512,4,3>: persistent workgroups, 32-frame groups = whole 128-byte output lines, fp16 [bin][frame] image",
        "ms_per_launch": round(ms_fit, 4), "clips_per_s": round(clips / (ms_fit * 1e-3), 1),
        "algorithmic_bytes_per_launch": fit_bytes,
        "algorithmic_GBps": round(fit_bytes / (ms_fit * 1e-3) / 1e9, 1), "frac_hbm_peak": round(fit_bytes / (ms_fit * 1e-3) / 1e9 / 8000.0, 4),
        "frac_hbm_achievable": round(fit_bytes / (ms_fit * 1e-3) / 1e9 / 6300.0, 4)}
    del fit
    if cpu_clips > 0:
        import oracle
        oracle.set_num_threads(host_cores())                   # the CPU share this process really has
        host = a[:cpu_clips].cpu().numpy()
        oracle.stft_mag(host[:2], n_fft, hop, True)
        t0 = time.perf_counter()
        oracle.stft_mag(host, n_fft, hop, True)
        el = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": round(cpu_clips / el, 1), "unit": "clips/s", "cores": oracle.num_threads(),
                               "kind": "port", "sample": f"{cpu_clips} of the same clips ({el:.2f} s), oracle/adn_oracle.c "
                                                         "(float64 radix-2 FFT, OpenMP)"}
    del a, out
    torch.cuda.empty_cache()
    return res


def bench_f16(sd_np, dev, batch=256, steps=10, warmup=2, single_clip=True):
    """BASELINE configs[4] on one GPU: fp16 storage + fp16 MFMA forward at batch 256 (+ per-clip loss, as the main step)."""
    from audiodenoiser_amd import _lib
    from audiodenoiser_amd.loss import perceptual_loss_per_clip
    from audiodenoiser_amd.roofline import PEAK_MFMA_F16_TFLOPS
    net = make_net(sd_np, dev, "f16")
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.rand((batch, 1, F_BINS, T_FRAMES), generator=g, device=dev) * 4.0
    target = torch.rand((batch, 1, F_BINS, T_FRAMES), generator=g, device=dev) * 4.0
    L = _lib.load()
    with torch.no_grad():
        for _ in range(warmup):
            loss = perceptual_loss_per_clip(net(x), target)
        _lib.check(L.adn_unet_set_timing(net._handle, steps), "adn_unet_set_timing")
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = perceptual_loss_per_clip(net(x), target)
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
    assert bool(torch.isfinite(loss).all())
    ms_mean = launch_timings(net, steps)
    res = {"workload": f"BASELINE configs[4]: batch={batch} synthetic 513x256 spectrograms, full U-Net forward with fp16 "
                       "storage + fp16 MFMA (fp32 accumulate) inside, fp32 at the boundary, + per-clip perceptual loss",
           "steps": steps, "warmup": warmup, "dtype": "f16", "ms_per_step": round(el / steps * 1e3, 3),
           "frames_per_s": round(batch * T_FRAMES * steps / el, 1),
           "roofline": conv_roofline(ms_mean, batch, "direct_f16", PEAK_MFMA_F16_TFLOPS,
                                     "conv16_f16 (LDS-DMA staged direct implicit GEMM on v_mfma_f32_16x16x32_f16, one persistent workgroup per CU; LDS-resident "
                                     "weights for the 64 -> 64 layers, first layer fused into down1's second conv): all 17 3x3 layers", "conv_mfma_f16"),
           "forward": forward_summary(ms_mean, batch, "direct_f16", PEAK_MFMA_F16_TFLOPS)}
    # one clip through the same handle (forward only, back-to-back calls, HIP events): configs[0]'s shape on the fp16 path
    # (--no-f16-b1 under the profiler: 55 single-clip forwards would dominate the per-family means of the PMC passes)
    x1 = x[:1].contiguous()
    if not single_clip:
        net._release()
        del net, x, target
        torch.cuda.empty_cache()
        return res
    with torch.no_grad():
        for _ in range(5):
            net(x1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            net(x1)
        e1.record()
        torch.cuda.synchronize(dev)
    res["b1_ms_per_forward"] = round(e0.elapsed_time(e1) / 50, 4)
    net._release()
    del net, x, target
    torch.cuda.empty_cache()
    return res


def default_batch_per_gpu(world: int) -> int:
    """BASELINE configs[1] (batch 64 on one GPU) at N = 1; configs[3] (batch 2048 over 8 GPUs = 256 clips per GPU) on more
    than one rank -- weak scaling: the same 256 per GPU at 2 and 4 ranks."""
    return 64 if world <= 1 else 256


def timed_region(step, steps: int, warmup: int, sync, device, on_timed_start=None, gather_probe=None):
    """The contract's timing: `warmup` untimed steps, barrier + sync, EXACTLY `steps` steps, sync + barrier; the time is the
    MAX over ranks.  `step()` returns the gathered per-clip tensor of the step (rank order).  `sync()` waits for this rank's
    device work (torch.cuda.synchronize on a GPU, a no-op in the CPU test).  Also measured, outside the timed region: each
    rank's own time for the K steps (between its own sync points), how many ranks the collective really spans (every rank
    contributes its id to a gathered tensor) and the all-gather's own time (`gather_probe()` = one all-gather of the step's
    payload, timed over 20 calls).  Used by main() and, with a stub step on gloo, by tests/test_host_logic.py."""
    from audiodenoiser_amd import distributed as D
    rank = D.rank()
    allv = None
    for _ in range(warmup):
        allv = step()
    if on_timed_start is not None:
        on_timed_start()
    D.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        allv = step()
    sync()
    own = time.perf_counter() - t0
    D.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = D.max_over_ranks(elapsed, device)
    per_rank = D.gather_per_clip(torch.tensor([float(rank), own], dtype=torch.float64, device=device)).reshape(-1, 2).cpu()
    ids = sorted(int(v) for v in per_rank[:, 0].tolist())
    own_ms = (per_rank[:, 1] / steps * 1e3).tolist()
    gather_ms = None
    if gather_probe is not None:
        for _ in range(3):
            gather_probe()
        sync()
        D.barrier()
        g0 = time.perf_counter()
        for _ in range(20):
            gather_probe()
        sync()
        gather_ms = D.max_over_ranks((time.perf_counter() - g0) / 20 * 1e3, device)
    return {"elapsed_s": elapsed, "last": allv,
            "ranks": {"ranks_seen": len(set(ids)), "rank_ids": ids, "world_size": D.world_size(),
                      "rank_ms_per_step_min": round(min(own_ms), 3), "rank_ms_per_step_max": round(max(own_ms), 3),
                      "allgather_ms": None if gather_ms is None else round(gather_ms, 4),
                      "allgather_payload_floats_per_rank": None}}


def assemble_line(elapsed_s: float, steps: int, warmup: int, world: int, batch_per_gpu: int, dtype: str, ranks: dict,
                  extra_config=None) -> dict:
    """The contract's JSON line (without the roofline / sub-benchmark objects main() adds on a GPU)."""
    f16 = dtype == "f16"
    frames = batch_per_gpu * world * T_FRAMES * steps
    cfg = ("BASELINE configs[3]: batch=%d synthetic 513x256 fp32 spectrograms sharded over %d MI355X (%d clips per GPU)"
           % (batch_per_gpu * world, world, batch_per_gpu)) if world > 1 else \
          ("BASELINE configs[1]: batch=%d synthetic 513x256 fp32 spectrograms on one MI355X" % batch_per_gpu)
    config = {"workload": cfg + ", full U-Net forward "
                          + ("[fp16 storage + fp16 MFMA inside, BASELINE configs[4]] " if f16 else "")
                          + "+ per-clip perceptual loss" + (" + RCCL all-gather of the per-clip values" if world > 1 else ""),
              "batch_per_gpu": batch_per_gpu, "global_batch": batch_per_gpu * world, "freq_bins": F_BINS, "frames": T_FRAMES,
              "parallelism": f"clips sharded over {world} rank(s), weights replicated, no data-path collective",
              "timed_region_s": round(elapsed_s, 3)}
    if extra_config:
        config.update(extra_config)
    return {"metric": "spectrogram frames/sec (forward), 513x256 fp32",
            "value": round(frames / elapsed_s, 1),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": round(elapsed_s / steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": dtype,
            "data": "synthetic",
            "config": config,
            "ranks": ranks}


def kernel_names(f16: bool):
    """(algo, peak, kernel description, PMC traffic key, Winograd tile mode) of the dominant 3x3 kernel of this process's
    handles (ADN_CONV_ALGO / ADN_WINO_TILE are read by libadn when a handle is created)."""
    from audiodenoiser_amd.roofline import PEAK_MFMA_F16_TFLOPS, PEAK_MFMA_F32_TFLOPS
    direct = f16 or not bool(net_uses_winograd())
    algo = "direct_f16" if f16 else ("direct" if direct else "winograd")
    peak = PEAK_MFMA_F16_TFLOPS if f16 else PEAK_MFMA_F32_TFLOPS
    wmode = wino_tile_mode()
    kname = ("conv16_f16 (LDS-DMA staged direct implicit GEMM, fp16 MFMA v_mfma_f32_16x16x32_f16)" if f16 else
             "conv_mfma<float> (direct implicit GEMM, fp32 MFMA)" if direct else
             "wino_conv_dma_f32 (Winograd F(2x2,3x3), fp32 MFMA v_mfma_f32_16x16x4_f32)" if wmode == "2" else
             "wino4_conv_f32 (Winograd F(4x4,3x3), fp32 MFMA v_mfma_f32_16x16x4_f32)")
    tkey = ("conv_mfma_f16" if f16 else "conv_mfma_f32" if direct else
            "wino_conv_dma_f32" if wmode == "2" else "wino4_conv_f32")
    return algo, peak, kname, tkey, wmode


def bench_fp32_b256(sd_np, dev, batch=256, steps=10, warmup=2):
    """The headline step at north_star's batch (256 clips = one rank's shard of BASELINE configs[3]) on this one GPU."""
    from audiodenoiser_amd import _lib
    from audiodenoiser_amd.loss import perceptual_loss_per_clip
    net = make_net(sd_np, dev, "f32")
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.rand((batch, 1, F_BINS, T_FRAMES), generator=g, device=dev) * 4.0
    target = torch.rand((batch, 1, F_BINS, T_FRAMES), generator=g, device=dev) * 4.0
    L = _lib.load()
    with torch.no_grad():
        for _ in range(warmup):
            loss = perceptual_loss_per_clip(net(x), target)
        _lib.check(L.adn_unet_set_timing(net._handle, steps), "adn_unet_set_timing")
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = perceptual_loss_per_clip(net(x), target)
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
    assert bool(torch.isfinite(loss).all())
    ms_mean = launch_timings(net, steps)
    algo, peak, kname, tkey, wmode = kernel_names(False)
    roof = conv_roofline(ms_mean, batch, algo, peak, kname, tkey, wmode)
    res = {"workload": f"batch={batch} synthetic 513x256 fp32 spectrograms on one MI355X (north_star's batch; the per-GPU "
                       "shard of BASELINE configs[3]), full U-Net forward + per-clip perceptual loss",
           "steps": steps, "warmup": warmup, "dtype": "f32", "ms_per_step": round(el / steps * 1e3, 3),
           "value": round(batch * T_FRAMES * steps / el, 1), "unit": "frames/s",
           "frac": roof["frac"], "roofline": {k: roof[k] for k in ("bound", "achieved", "peak", "unit", "frac", "kernel", "avg_launch_ms")},
           "forward_kernel_ms": round(float(ms_mean.sum()), 3)}
    net._release()
    del net, x, target
    torch.cuda.empty_cache()
    return res


def bench_exact_f32(sd_np, dev, batch=64, steps=20, warmup=3):
    """The headline step with exact-fp32 matrix instructions EVERYWHERE (ADN_CONVT_SPLIT=0 when the handle is created: the transposed
    convolutions as v_mfma_f32_32x32x2_f32 GEMMs instead of the three-term bf16 split): the number to quote if the split -- an
    fp32-accurate product formed on the bf16 matrix cores -- is not accepted as fp32 arithmetic."""
    from audiodenoiser_amd.loss import perceptual_loss_per_clip
    saved = os.environ.get("ADN_CONVT_SPLIT")
    os.environ["ADN_CONVT_SPLIT"] = "0"
    try:
        net = make_net(sd_np, dev, "f32")
        g = torch.Generator(device=dev).manual_seed(0)
        x = torch.rand((batch, 1, F_BINS, T_FRAMES), generator=g, device=dev) * 4.0
        target = torch.rand((batch, 1, F_BINS, T_FRAMES), generator=g, device=dev) * 4.0
        with torch.no_grad():
            for _ in range(warmup):
                loss = perceptual_loss_per_clip(net(x), target)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(steps):
                loss = perceptual_loss_per_clip(net(x), target)
            torch.cuda.synchronize(dev)
            el = time.perf_counter() - t0
        assert bool(torch.isfinite(loss).all())
        net._release()
        del net, x, target
        torch.cuda.empty_cache()
    finally:
        if saved is None:
            os.environ.pop("ADN_CONVT_SPLIT", None)
        else:
            os.environ["ADN_CONVT_SPLIT"] = saved
    return {"what": f"batch={batch} x 513x256, the same step with ADN_CONVT_SPLIT=0: every matrix instruction is an exact-fp32 MFMA",
            "steps": steps, "warmup": warmup, "ms_per_step": round(el / steps * 1e3, 3),
            "value": round(batch * T_FRAMES * steps / el, 1), "unit": "frames/s"}


def bench_latency(sd_np, dev, iters=30):
    """Single-clip / small-batch latency: BASELINE configs[0] is one clip through the forward (test.py:112-113 runs its whole
    test set -- 5 clips of 257x188 -- as one batch; train.py:84 validates at batch 16 of 256x64).  Default kernels, and the
    serving switches libadn reads when a handle is created (ADN_WINO_TILE=2 + ADN_WINO_SPLITK=1: finer F(2x2,3x3) grid and
    split-K, which fill the chip at one clip; they change the summation order, so they are opt-in)."""
    shapes = (("513x256_b1", 1, F_BINS, T_FRAMES), ("257x188_b1", 1, 257, 188), ("257x188_b5_test_py", 5, 257, 188),
              ("256x64_b16_train_py", 16, 256, 64))

    def run(env):
        saved = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            net = make_net(sd_np, dev, "f32")
            out = {}
            with torch.no_grad():
                for name, b, f, t in shapes:
                    x = torch.rand((b, 1, f, t), device=dev) * 4.0
                    for _ in range(3):
                        net(x)
                    reps = []                                  # median of three timed groups: one stall of the box does not make the figure
                    for _ in range(3):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        torch.cuda.synchronize(dev)
                        e0.record()
                        for _ in range(iters):
                            y = net(x)
                        e1.record()
                        torch.cuda.synchronize(dev)
                        reps.append(e0.elapsed_time(e1) / iters)
                    assert bool(torch.isfinite(y).all())
                    ms = sorted(reps)[1]
                    out[name] = {"ms_per_forward": round(ms, 4), "frames_per_s": round(b * t / (ms * 1e-3), 1)}
            net._release()
            return out
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v

    res = {"what": "forward only (no loss), back-to-back calls on one stream, HIP events, median of three groups of 30; default = no "
                   "environment variable set: per launch, F(4x4,3x3) where its grid fills the chip, else the cheaper of F(2x2,3x3) / "
                   "F(4x4,3x3) cut along K; batch_invariant = "
                   "ADN_BATCH_INVARIANT=1 (one kernel per layer by geometry: bit-equal clips across batch sizes; the default of "
                   "rounds 1-3); serving = ADN_WINO_TILE=2 ADN_WINO_SPLITK=1 (F(2x2,3x3) + split-K everywhere)",
           "default": run({}), "batch_invariant": run({"ADN_BATCH_INVARIANT": "1"}),
           "serving": run({"ADN_WINO_TILE": "2", "ADN_WINO_SPLITK": "1"})}
    torch.cuda.empty_cache()
    return res


def bench_e2e_config0(sd_np, dev, iters=30, with_cpu=True):
    """BASELINE configs[0] end to end as ONE stream sequence (the reference's test.py:94-113 flow with create_test_dataset.py:35-41
    in front): resident audio (3 s @ 44.1 kHz) -> adn_stft_mag_fit (STFT 1024/256 centred + the loader's fp16 round trip and
    crop to 513x256) -> adn_unet_forward, at batch 1 and 64, HIP-event timed; the oracle chain (C STFT + loader rule + the
    torch forward) timed on the host beside it."""
    from audiodenoiser_amd.stft import prepare, stft_magnitude_fit
    net = make_net(sd_np, dev, "f32")
    prepare(dev, 1024)
    out = {"what": "audio resident in HBM -> adn_stft_mag_fit -> adn_unet_forward on one stream (no host round trip), HIP events "
                   "around the whole sequence; 132 300-sample clips, n_fft 1024, hop 256, centred, target 513x256"}
    g = torch.Generator(device=dev).manual_seed(0)
    with torch.no_grad():
        for b in (1, 64):
            audio = torch.rand((b, 132300), generator=g, device=dev) * 2 - 1
            for _ in range(3):
                y = net(stft_magnitude_fit(audio, (F_BINS, T_FRAMES), 1024, 256, True))
            n_it = iters if b == 1 else max(5, iters // 3)
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            torch.cuda.synchronize(dev)
            e0.record()
            for _ in range(n_it):
                x = stft_magnitude_fit(audio, (F_BINS, T_FRAMES), 1024, 256, True)
            e1.record()
            for _ in range(n_it):
                y = net(stft_magnitude_fit(audio, (F_BINS, T_FRAMES), 1024, 256, True))
            e2.record()
            torch.cuda.synchronize(dev)
            assert bool(torch.isfinite(y).all())
            ms = e1.elapsed_time(e2) / n_it
            out[f"b{b}"] = {"ms_per_clip_batch": round(ms, 4), "stft_fit_ms": round(e0.elapsed_time(e1) / n_it, 4),
                            "clips_per_s": round(b / (ms * 1e-3), 1), "frames_per_s": round(b * T_FRAMES / (ms * 1e-3), 1)}
            del audio, x, y
    net._release()
    if with_cpu:
        import oracle
        from oracle import unet_torch
        share = host_cores()
        oracle.set_num_threads(share)
        torch.set_num_threads(min(share, 16))
        sd = unet_torch.to_torch_state(sd_np)
        clips = np.random.default_rng(0).uniform(-1, 1, (2, 132300)).astype(np.float32)

        def chain(a):
            mag = oracle.stft_mag(a, 1024, 256, True)
            xs = np.stack([oracle.quantize_pad(m, (F_BINS, T_FRAMES)) for m in (mag if mag.ndim == 3 else mag[None])])[:, None]
            return unet_torch.unet_forward(sd, torch.from_numpy(xs))
        chain(clips[:1])
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 6.0 and n < 8:
            chain(clips[:1])
            n += 1
        el = (time.perf_counter() - t0) / n
        out["cpu_baseline"] = {"value": round(1.0 / el, 3), "unit": "clips/s", "ms_per_clip": round(el * 1e3, 1), "cores": min(share, 16),
                               "kind": "port", "sample": f"{n} single clips through oracle.stft_mag (C, OpenMP) + quantize_pad + "
                                                         "oracle/unet_torch.py (oneDNN)"}
    torch.cuda.empty_cache()
    return out


def bench_host_boundary(sd_np, dev, batch=64, iters=8):
    """The PCIe-inclusive rate of the headline workload (never `value`): the reference's own call shape hands the model CPU tensors
    (test.py:100-113), so a drop-in caller that keeps its spectrograms in host memory pays the two copies.  Three forms, batch 64 x
    513x256 fp32 (33.6 MB each way): (a) `model(x_cpu)` exactly as test.py calls it (pageable memory, wall clock), (b) pinned buffers,
    copy in -> forward -> copy out on one stream, (c) the same with the copies on their own streams, double buffered, so that batch
    i+1 goes in and batch i-1 comes out under batch i's forward (batch i+1's copy in submitted before batch i's copy out).  (c') the packaged
    form of (c), `UNet.forward_host_batches`, on pinned inputs / output views and on pageable inputs / copied outputs.  Plus (d) audio in host memory -> adn_stft_mag_fit -> forward ->
    magnitudes out (configs[0]'s flow; 33.9 MB of audio per 64 clips)."""
    from audiodenoiser_amd.stft import prepare, stft_magnitude_fit
    net = make_net(sd_np, dev, "f32")
    prepare(dev, 1024)
    shape = (batch, 1, F_BINS, T_FRAMES)
    gen = torch.Generator().manual_seed(0)
    x_page = torch.rand(shape, generator=gen) * 4.0
    x_pin = x_page.clone().pin_memory()
    y_pin = [torch.empty(shape).pin_memory() for _ in range(2)]
    a_pin = (torch.rand((batch, 132300), generator=gen) * 2 - 1).pin_memory()
    mb = x_page.numel() * 4 / 1e6
    out = {"what": f"batch {batch} x {F_BINS}x{T_FRAMES} fp32 with the spectrograms in HOST memory ({mb:.1f} MB in, {mb:.1f} MB out per "
                   "batch): what a caller that keeps the reference's CPU-tensor call shape gets; `value` of this line is measured with "
                   "resident inputs"}

    def rate(ms):
        return {"ms_per_batch": round(ms, 3), "frames_per_s": round(batch * T_FRAMES / (ms * 1e-3), 1)}
    with torch.no_grad():
        for _ in range(2):
            y = net(x_page)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(iters):
            y = net(x_page)
        assert not y.is_cuda
        out["reference_call_shape_pageable"] = rate((time.perf_counter() - t0) / iters * 1e3)

        comp = torch.cuda.current_stream(dev)
        xd = [torch.empty(shape, device=dev) for _ in range(2)]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for timed in (False, True):
            if timed:
                torch.cuda.synchronize(dev)
                e0.record()
            for _ in range(iters if timed else 2):
                xd[0].copy_(x_pin, non_blocking=True)
                y = net(xd[0])
                y_pin[0].copy_(y, non_blocking=True)
            e1.record()
        torch.cuda.synchronize(dev)
        out["pinned_one_stream"] = rate(e0.elapsed_time(e1) / iters)

        s_in, s_out = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        ev_in = [torch.cuda.Event() for _ in range(2)]
        ev_comp = [torch.cuda.Event() for _ in range(2)]
        n_it = 2 * iters
        fwd_ev = []

        def copy_in(i):
            k = i & 1
            with torch.cuda.stream(s_in):
                if i >= 2:
                    s_in.wait_event(ev_comp[k])            # batch i-2's forward has read xd[k]
                xd[k].copy_(x_pin, non_blocking=True)
                ev_in[k].record(s_in)
        torch.cuda.synchronize(dev)
        copy_in(0)
        for i in range(n_it + 2):
            k = i & 1
            # the next batch goes in BEFORE this batch's copy out is submitted: the copy queue is served in submission order, and a
            # copy out that waits for its forward holds up every copy submitted behind it (tools/host_boundary_timeline.py)
            copy_in(i + 1)
            if i == 2:
                e0.record()                                # behind the forward of batch 1: the pipeline is full, the clocks are up
            comp.wait_event(ev_in[k])
            fe = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            fe[0].record()
            y = net(xd[k])
            fe[1].record()
            if i >= 2:
                fwd_ev.append(fe)
            ev_comp[k].record(comp)
            with torch.cuda.stream(s_out):
                s_out.wait_event(ev_comp[k])
                y_pin[k].copy_(y, non_blocking=True)
                y.record_stream(s_out)
        comp.wait_stream(s_out)
        e1.record()
        comp.wait_stream(s_in)
        torch.cuda.synchronize(dev)
        out["pinned_copy_streams_double_buffered"] = rate(e0.elapsed_time(e1) / n_it)
        out["pinned_copy_streams_double_buffered"]["forward_ms_inside"] = round(sum(a.elapsed_time(b) for a, b in fwd_ev) / len(fwd_ev), 3)
        assert bool(torch.isfinite(y_pin[0]).all()) and bool(torch.isfinite(y_pin[1]).all())

        # the packaged form of (c): UNet.forward_host_batches, wall clock over the whole generator (first batch's fill included)
        for name, src, cp in (("forward_host_batches_pinned_views", x_pin, False), ("forward_host_batches_pageable_copies", x_page, True)):
            for timed in (False, True):
                n_b = n_it if timed else 3
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                got = sum(1 for _ in net.forward_host_batches((src for _ in range(n_b)), copy=cp))
                el = time.perf_counter() - t0
            assert got == n_it
            out[name] = rate(el / n_it * 1e3)

        ad = torch.empty(a_pin.shape, device=dev)
        for timed in (False, True):
            if timed:
                torch.cuda.synchronize(dev)
                e0.record()
            for _ in range(iters if timed else 2):
                ad.copy_(a_pin, non_blocking=True)
                y = net(stft_magnitude_fit(ad, (F_BINS, T_FRAMES), 1024, 256, True))
                y_pin[0].copy_(y, non_blocking=True)
            e1.record()
        torch.cuda.synchronize(dev)
        out["audio_in_pinned_one_stream"] = rate(e0.elapsed_time(e1) / iters)
    del xd, ad, y
    net._release()
    torch.cuda.empty_cache()
    return out


def per_gpu_reference():
    """fp32_b256 of the newest recorded N = 1 line (profiles/rNN_bench.json) that was taken with this very build: the one-GPU
    figure at 256 clips per GPU that an N > 1 value is to be divided by (the N = 1 headline runs 64 clips)."""
    import glob
    digest = lib_digest()[:12]
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench.json")), reverse=True)
    for path in paths:
        try:
            with open(path) as fh:
                rec = json.load(fh)
        except (OSError, ValueError):
            continue
        if rec.get("config", {}).get("lib_digest") == digest and "fp32_b256" in rec:
            name = os.path.relpath(path, ROOT)
            return {"value": rec["fp32_b256"]["value"], "unit": "frames/s", "batch_per_gpu": 256,
                    "source": f"{name} fp32_b256 (N = 1, same library digest)"}
    return {"value": None, "note": "no recorded N = 1 line of this build with fp32_b256 under profiles/ (rNN_bench.json)"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=160, help="timed steps (default 160: >= 5 s of timed region at 35 ms per step)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch-per-gpu", type=int, default=None,
                    help="default: 64 on one GPU (BASELINE configs[1]), 256 on more than one rank (configs[3] = 2048 / 8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the sub-benchmarks (stft = configs[2], f16 = configs[4], fp32_b256, b1)")
    ap.add_argument("--extras", default="stft,f16,fp32_b256,b1,exact_f32,e2e_config0,host_boundary",
                    help="which sub-benchmarks to run (comma list; tools/profile_bench.sh profiles with stft,f16 only, so that the "
                         "kernel statistics of the headline kernel are not mixed with other batch sizes)")
    ap.add_argument("--stft-steps", type=int, default=20)
    ap.add_argument("--f16-steps", type=int, default=10)
    ap.add_argument("--no-f16-b1", action="store_true", help="skip the single-clip timing inside the f16 extra (profiler runs)")
    ap.add_argument("--b256-steps", type=int, default=10)
    ap.add_argument("--no-stft-cpu", action="store_true", help="skip the C-oracle STFT timing inside the stft sub-benchmark")
    ap.add_argument("--no-finite-check", action="store_true",
                    help="skip the finite-output assertion (work-in-progress kernels)")
    ap.add_argument("--dtype", choices=("f32", "f16"), default="f32",
                    help="f32 = headline metric (default); f16 = run the MAIN loop on the fp16 path (BASELINE configs[4])")
    args = ap.parse_args()

    from audiodenoiser_amd import _lib
    from audiodenoiser_amd import distributed as D
    from audiodenoiser_amd.loss import perceptual_loss_per_clip
    from audiodenoiser_amd.weights import make_state_dict

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device (no CPU path)")
    if args.steps < 1 or args.steps > 4096:
        raise SystemExit("--steps must be in [1, 4096]")
    # ADN_BENCH_REHEARSAL=1 (one-GPU boxes only): the N > 1 code path with every rank on device 0 and gloo moving the CUDA
    # tensors -- rehearses the driver's torch.distributed.run command line where no second GPU exists; never a measurement
    rehearsal = os.environ.get("ADN_BENCH_REHEARSAL", "") not in ("", "0")
    rank, local_rank, world = D.init_from_env("gloo" if rehearsal else "nccl")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    dev = torch.device("cuda", (local_rank % torch.cuda.device_count() if rehearsal else local_rank) if world > 1 else 0)
    torch.cuda.set_device(dev)

    sd_np = make_state_dict(1234)                                     # replicated weights, regenerated per rank
    net = make_net(sd_np, dev, args.dtype)
    f16 = args.dtype == "f16"

    b = args.batch_per_gpu if args.batch_per_gpu else default_batch_per_gpu(world)
    g = torch.Generator(device=dev).manual_seed(rank)                 # per-rank shard of the global batch
    x = torch.rand((b, 1, F_BINS, T_FRAMES), generator=g, device=dev) * 4.0      # resident in HBM
    target = torch.rand((b, 1, F_BINS, T_FRAMES), generator=g, device=dev) * 4.0

    def step():
        y = net(x)
        loss = perceptual_loss_per_clip(y, target)            # (b, 4): total, stft, mel, l1 per clip
        return D.gather_per_clip(loss.reshape(-1))

    L = _lib.load()
    payload = torch.zeros(4 * b, dtype=torch.float32, device=dev)
    with torch.no_grad():
        tr = timed_region(step, args.steps, args.warmup, lambda: torch.cuda.synchronize(dev), dev,
                          on_timed_start=lambda: _lib.check(L.adn_unet_set_timing(net._handle, args.steps), "adn_unet_set_timing"),
                          gather_probe=(lambda: D.gather_per_clip(payload)) if world > 1 else None)
    elapsed, allv = tr["elapsed_s"], tr["last"]
    tr["ranks"]["allgather_payload_floats_per_rank"] = 4 * b
    assert allv.numel() == 4 * b * world and (args.no_finite_check or bool(torch.isfinite(allv).all()))
    assert tr["ranks"]["ranks_seen"] == world, tr["ranks"]
    ms_mean = launch_timings(net, args.steps)      # per-launch durations of the timed steps (events inside libadn)

    if rank == 0:
        algo, peak, kname, tkey, wmode = kernel_names(f16)
        arith = ("fp16 storage + fp16 MFMA, fp32 accumulate" if f16 else
                 "fp32 storage; 3x3 layers on exact-fp32 MFMA (Winograd F(4x4,3x3)); transposed convolutions "
                 + ("with both fp32 operands split into three bf16 terms, six bf16-MFMA products, fp32 accumulate (fp32-accurate: "
                    "whole-network parity 5.5e-6 of max|y| either way; `exact_f32` = the same step with exact-fp32 MFMA everywhere)"
                    if convt_uses_split_bf16(False) else "on exact-fp32 MFMA (ADN_CONVT_SPLIT=0)"))
        out = assemble_line(elapsed, args.steps, args.warmup, world, b, args.dtype, tr["ranks"],
                            {"lib_digest": lib_digest()[:12], "arithmetic": arith})
        out["roofline"] = conv_roofline(ms_mean, b, algo, peak, kname, tkey, wmode)
        out["forward"] = forward_summary(ms_mean, b, algo, peak, wmode)
        if rehearsal:
            out["data"] = "synthetic; REHEARSAL of the N > 1 path on one GPU (gloo, every rank on device 0): not a measurement"
        if world > 1:
            out["config"]["per_gpu_reference"] = per_gpu_reference()
            out["config"]["compare_with"] = ("the N = 1 line's fp32_b256.value (the same 256 clips per GPU on one GPU); the N = 1 "
                                             "headline value is BASELINE configs[1] at 64 clips")
        del allv
        if world == 1 and not args.no_extras:
            net._workspace = None
            del x, target
            torch.cuda.empty_cache()
            extras = {e.strip() for e in args.extras.split(",") if e.strip()}
            if "stft" in extras:
                out["stft"] = bench_stft(dev, steps=args.stft_steps, cpu_clips=0 if args.no_stft_cpu else 4096)
            if not f16:
                if "f16" in extras:
                    out["f16"] = bench_f16(sd_np, dev, steps=args.f16_steps, single_clip=not args.no_f16_b1)
                if "fp32_b256" in extras:
                    out["fp32_b256"] = bench_fp32_b256(sd_np, dev, steps=args.b256_steps)
                if "b1" in extras:
                    out["b1"] = bench_latency(sd_np, dev)
                if "exact_f32" in extras and convt_uses_split_bf16(False):
                    out["exact_f32"] = bench_exact_f32(sd_np, dev)
                if "e2e_config0" in extras:
                    out["e2e_config0"] = bench_e2e_config0(sd_np, dev, with_cpu=not args.no_cpu_baseline)
                if "host_boundary" in extras:
                    out["host_boundary"] = bench_host_boundary(sd_np, dev)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(sd_np)
        print(json.dumps(out), flush=True)
    D.barrier()
    if world > 1:
        torch.distributed.destroy_process_group()


def wino_tile_mode() -> str:
    """ADN_WINO_TILE as libadn reads it when a handle is created: "2" = F(2x2,3x3) for every 3x3 layer, "4" = F(4x4,3x3)
    for every plain 3x3 layer, anything else = per layer by tile fit (audiodenoiser_amd.roofline.winograd_tile)."""
    v = os.environ.get("ADN_WINO_TILE", "")
    return v if v in ("2", "4") else "auto"


def net_uses_winograd() -> bool:
    """The fp32 3x3 layers run the Winograd kernel unless ADN_CONV_ALGO=direct was set when the handle was created."""
    return os.environ.get("ADN_CONV_ALGO") != "direct"


if __name__ == "__main__":
    main()
