#!/usr/bin/env python3
"""BASELINE configs[2]: STFT magnitude only — n_clips x 3 s @ 44.1 kHz synthetic clips, n_fft 1024, hop 256,
centred.  Prints one JSON line with clips/s, frames/s and the HBM-roofline fraction
(algorithmic bytes = audio read once + magnitudes written once = 1 590 084 B per clip)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=10000)
    ap.add_argument("--length", type=int, default=132300)
    ap.add_argument("--n-fft", type=int, default=1024)
    ap.add_argument("--hop", type=int, default=256)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cpu-clips", type=int, default=64)
    ap.add_argument("--fit", action="store_true", help="also time adn_stft_mag_fit (the same clips -> (clips,1,513,256))")
    args = ap.parse_args()
    from audiodenoiser_amd import _lib
    from audiodenoiser_amd.stft import stft_n_frames
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    a = torch.rand((args.clips, args.length), generator=g, device=dev) * 2 - 1
    nfr = stft_n_frames(args.length, args.n_fft, args.hop, True)
    nb = args.n_fft // 2 + 1
    out = torch.empty((args.clips, nb, nfr), dtype=torch.float32, device=dev)
    L = _lib.load()
    st = torch.cuda.current_stream(dev).cuda_stream

    def run():
        _lib.check(L.adn_stft_mag(a.data_ptr(), args.clips, args.length, args.n_fft, args.hop, 1, out.data_ptr(), st),
                   "adn_stft_mag")
    for _ in range(args.warmup):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(args.steps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.steps
    bytes_per_clip = args.length * 4 + nb * nfr * 4
    gbs = args.clips * bytes_per_clip / (ms * 1e-3) / 1e9
    res = {"metric": "STFT magnitude clips/s", "clips": args.clips, "n_fft": args.n_fft, "hop": args.hop,
           "ms_per_launch": round(ms, 4), "clips_per_s": round(args.clips / (ms * 1e-3), 1),
           "frames_per_s": round(args.clips * nfr / (ms * 1e-3), 1),
           "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s",
                        "frac": round(gbs / 8000.0, 4), "bytes_per_clip": bytes_per_clip}}
    if args.fit:
        H, W = args.n_fft // 2 + 1, 256
        fit = torch.empty((args.clips, 1, H, W), dtype=torch.float32, device=dev)

        def run_fit():
            _lib.check(L.adn_stft_mag_fit(a.data_ptr(), args.clips, args.length, args.n_fft, args.hop, 1, fit.data_ptr(), H, W, st),
                       "adn_stft_mag_fit")
        for _ in range(args.warmup):
            run_fit()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(args.steps):
            run_fit()
        e1.record()
        torch.cuda.synchronize()
        ms_fit = e0.elapsed_time(e1) / args.steps
        fit_bytes = args.clips * ((W - 1) * args.hop + args.n_fft // 2 + H * W) * 4
        res["fit"] = {"ms_per_launch": round(ms_fit, 4), "algorithmic_GBps": round(fit_bytes / (ms_fit * 1e-3) / 1e9, 1),
                      "frac": round(fit_bytes / (ms_fit * 1e-3) / 1e9 / 8000.0, 4), "bytes": fit_bytes}
    if args.cpu_clips > 0:
        import oracle
        host = a[:args.cpu_clips].cpu().numpy()
        oracle.stft_mag(host[:2], args.n_fft, args.hop, True)
        t0 = time.perf_counter()
        oracle.stft_mag(host, args.n_fft, args.hop, True)
        el = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": round(args.cpu_clips / el, 1), "unit": "clips/s", "cores": oracle.num_threads(),
                               "kind": "port", "sample": f"{args.cpu_clips} clips, oracle/adn_oracle.c (float64 radix-2 FFT, OpenMP)"}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
