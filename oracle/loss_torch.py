"""torch restatement of the reference's CombinedPerceptualLoss, per clip — TEST INFRASTRUCTURE (oracle/__init__.py).

Follows ``/root/reference/code/loss.py``:
* ``MultiScaleSTFTLoss`` (``:6-35``): mean over the frequency axis -> (B, T) series; for (n_fft, hop) in
  ((63,16),(32,8),(16,4)): ``torch.stft(x, n_fft, hop_length=hop, window=ones(n_fft), pad_mode="constant",
  return_complex=True)`` (center=True), ``abs``, ``F.l1_loss``; average of the three.
* ``MelSpectrogramLoss`` (``:37-69``): ``torchaudio.transforms.MelSpectrogram(sample_rate=8000, n_fft=63,
  hop_length=16, n_mels=64)`` per sample, ``l1_loss``.  torchaudio is NOT installed here, so this part restates
  its published defaults — periodic Hann window, centre reflect padding, power 2, HTK mel scale, no filter
  normalisation, f_min 0, f_max sr/2 (``torchaudio.functional.melscale_fbanks``) — and is **parity unpinned**.
* ``CombinedPerceptualLoss`` (``:71-95``): 0.4 stft + 0.4 mel + 0.2 L1.

Per-clip values: every clip contributes the same number of elements to each ``l1_loss``, so the batch losses of
the reference equal the means over clips of the values returned here (checked in tests/test_loss_oracle.py for
the parts that only need torch).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

SCALES = ((63, 16), (32, 8), (16, 4))
MEL_SR, MEL_NFFT, MEL_HOP, MEL_NMELS = 8000, 63, 16, 64
W_STFT, W_MEL, W_L1 = 0.4, 0.4, 0.2


def mel_filterbank(n_freqs: int = MEL_NFFT // 2 + 1, n_mels: int = MEL_NMELS, sample_rate: int = MEL_SR) -> np.ndarray:
    """(n_freqs, n_mels) triangular filters, HTK scale, norm=None, f_min=0, f_max=sample_rate/2 (float64)."""
    def hz2mel(f):
        return 2595.0 * math.log10(1.0 + f / 700.0)

    def mel2hz(m):
        return 700.0 * (10.0 ** (m / 2595.0) - 1.0)

    all_freqs = np.linspace(0.0, sample_rate // 2, n_freqs)
    m_pts = np.linspace(hz2mel(0.0), hz2mel(sample_rate / 2.0), n_mels + 2)
    f_pts = np.array([mel2hz(m) for m in m_pts])
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return np.maximum(0.0, np.minimum(down, up))


@torch.no_grad()
def per_clip(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """(B,1,F,T) x2 -> (B,4) = [total, stft, mel, l1] per clip (float32, computed in float32 like the reference)."""
    b = pred.shape[0]
    p = pred.mean(dim=2).squeeze(1)
    q = target.mean(dim=2).squeeze(1)
    stft = torch.zeros(b, dtype=torch.float32)
    for n_fft, hop in SCALES:
        win = torch.ones(n_fft)
        pm = torch.stft(p, n_fft=n_fft, hop_length=hop, return_complex=True, pad_mode="constant", window=win).abs()
        qm = torch.stft(q, n_fft=n_fft, hop_length=hop, return_complex=True, pad_mode="constant", window=win).abs()
        stft += (pm - qm).abs().reshape(b, -1).mean(dim=1)
    stft /= len(SCALES)
    fb = torch.from_numpy(mel_filterbank()).float()
    win = torch.hann_window(MEL_NFFT, periodic=True)

    def mel(x):
        s = torch.stft(x, n_fft=MEL_NFFT, hop_length=MEL_HOP, window=win, center=True, pad_mode="reflect",
                       return_complex=True).abs().pow(2.0)          # (B, 32, frames)
        return torch.matmul(s.transpose(1, 2), fb).transpose(1, 2)  # (B, 64, frames)
    melv = (mel(p) - mel(q)).abs().reshape(b, -1).mean(dim=1)
    l1 = (pred - target).abs().reshape(b, -1).mean(dim=1)
    total = W_STFT * stft + W_MEL * melv + W_L1 * l1
    return torch.stack([total, stft, melv, l1], dim=1)


@torch.no_grad()
def batch_reference_parts(pred: torch.Tensor, target: torch.Tensor):
    """The torch-only parts written exactly as the reference's batch code (loss.py:12-35, 86): (stft, l1)."""
    p = pred.mean(dim=2).squeeze(1)
    q = target.mean(dim=2).squeeze(1)
    loss = 0.0
    for n_fft, hop in SCALES:
        window = torch.ones(n_fft)
        pm = torch.abs(torch.stft(p, n_fft=n_fft, hop_length=hop, return_complex=True, pad_mode="constant", window=window))
        qm = torch.abs(torch.stft(q, n_fft=n_fft, hop_length=hop, return_complex=True, pad_mode="constant", window=window))
        loss = loss + F.l1_loss(pm, qm)
    return loss / len(SCALES), F.l1_loss(pred, target)
