#!/usr/bin/env python3
"""End-to-end mirror of the reference's test script (``/root/reference/code/test.py:74-176`` ``test_single_noise_type``)
on the MI355X path: load ``clean_{noise}.npy`` / ``noisy_{noise}.npy`` (N, F, T), run the U-Net on the device, compute
the four perceptual losses, reconstruct a few clips with Griffin-Lim and write them as wav files plus the metrics text
file.  Plots (matplotlib) and ``soundfile`` are replaced by ``audiodenoiser_amd.wav.write_wav``; nothing else differs.

    python tools/run_test_set.py --data ./data/test_processed --models ./saved_models --out ./data/test_output_ensemble

With ``--synthetic`` it fabricates a small test set and random-initialised checkpoints first (no reference data needed),
which is what the smoke run on the GPU box uses.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402

SAMPLE_RATE, N_FFT, HOP = 8000, 512, 128          # test.py:19-21
NOISE_TYPES = ["white", "urban", "reverb", "noise_cancellation"]   # test.py:24


def fabricate(data_dir, models_dir, n_clips=6):
    from audiodenoiser_amd.model import UNet
    from audiodenoiser_amd.stft import audio_to_spectrogram
    from audiodenoiser_amd.weights import make_state_dict
    os.makedirs(data_dir, exist_ok=True)
    os.makedirs(models_dir, exist_ok=True)
    rng = np.random.default_rng(0)
    sd = {k: torch.from_numpy(np.array(v)) for k, v in make_state_dict(1234).items()}
    for nt in NOISE_TYPES[:2]:
        clean = [0.3 * np.sin(2 * np.pi * rng.uniform(100, 900) * np.arange(24000) / SAMPLE_RATE).astype(np.float32)
                 for _ in range(n_clips)]
        noisy = [c + rng.normal(0, 0.05, c.shape).astype(np.float32) for c in clean]
        np.save(os.path.join(data_dir, f"clean_{nt}.npy"), np.stack([audio_to_spectrogram(c) for c in clean]))
        np.save(os.path.join(data_dir, f"noisy_{nt}.npy"), np.stack([audio_to_spectrogram(c) for c in noisy]))
        torch.save(sd, os.path.join(models_dir, f"unet_denoiser_{nt}.pth"))
    assert UNet  # imported for side-effect-free validation of the package


def run_noise_type(nt, data_dir, models_dir, out_dir, dev, n_audio=5):
    from audiodenoiser_amd.griffin_lim import griffin_lim_reconstruction
    from audiodenoiser_amd.loss import CombinedPerceptualLoss
    from audiodenoiser_amd.model import UNet
    from audiodenoiser_amd.wav import write_wav
    clean_path, noisy_path = (os.path.join(data_dir, f"{k}_{nt}.npy") for k in ("clean", "noisy"))
    model_path = os.path.join(models_dir, f"unet_denoiser_{nt}.pth")
    if not (os.path.exists(clean_path) and os.path.exists(noisy_path)):
        print(f"Skipping {nt}, missing {clean_path} or {noisy_path}")            # test.py:89-91
        return None
    if not os.path.exists(model_path):
        print(f"Model file not found: {model_path}")                             # test.py:59-60
        return None
    model = UNet(1, 1)
    model.load_state_dict(torch.load(model_path, map_location="cpu", weights_only=True))   # test.py:65
    model = model.to(dev).eval()
    clean = torch.from_numpy(np.load(clean_path).astype(np.float32)).unsqueeze(1).to(dev)
    noisy = torch.from_numpy(np.load(noisy_path).astype(np.float32)).unsqueeze(1).to(dev)
    print(f"Found {len(noisy)} test samples for noise type '{nt}'")
    t0 = time.perf_counter()
    with torch.no_grad():
        denoised = model(noisy)                                                  # test.py:112-113
        total, stft, mel, l1 = CombinedPerceptualLoss()(denoised, clean)         # test.py:118-122
    k = min(n_audio, len(noisy))
    noisy_audio = griffin_lim_reconstruction(noisy[:k, 0], N_FFT, HOP)           # test.py:103-109
    den_audio = griffin_lim_reconstruction(denoised[:k, 0], N_FFT, HOP)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for i in range(k):
        write_wav(os.path.join(out_dir, f"{nt}_noisy_{i}.wav"), noisy_audio[i].cpu().numpy(), SAMPLE_RATE)
        write_wav(os.path.join(out_dir, f"{nt}_denoised_{i}.wav"), den_audio[i].cpu().numpy(), SAMPLE_RATE)
    with open(os.path.join(out_dir, f"{nt}_metrics.txt"), "w") as fh:           # test.py:131-138
        fh.write(f"Perceptual metrics for noise type '{nt}':\n")
        for name, v in (("Total", total), ("STFT", stft), ("Mel", mel), ("L1", l1)):
            fh.write(f"{name} Loss: {float(v):.6f}\n")
            print(f"{name} Loss: {float(v):.6f}")
    print(f"[{nt}] forward + losses + {2 * k} Griffin-Lim reconstructions: {dt * 1e3:.1f} ms")
    return float(total)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", default="./data/test_processed")
    ap.add_argument("--models", default="./saved_models")
    ap.add_argument("--out", default="./data/test_output_ensemble")
    ap.add_argument("--synthetic", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    if args.synthetic:
        fabricate(args.data, args.models)
    os.makedirs(args.out, exist_ok=True)
    done = [nt for nt in NOISE_TYPES if run_noise_type(nt, args.data, args.models, args.out, dev) is not None]
    print(f"processed noise types: {done}")
    return 0 if done else 1


if __name__ == "__main__":
    sys.exit(main())
