"""numpy restatement of the reference's Griffin-Lim reconstruction — TEST INFRASTRUCTURE (oracle/__init__.py).

Follows ``/root/reference/code/test.py:29-48``:

    angles = exp(2j*pi*rand(F, T)); S = magnitude * angles
    repeat `iterations` times:  audio = librosa.istft(S, hop_length=hop)
                                Z = librosa.stft(audio, n_fft=n_fft, hop_length=hop)
                                S = abs(Z) * exp(1j * angle(Z))
    return librosa.istft(S, hop_length=hop)

Note what the loop does NOT do: the target magnitude is never re-imposed, so each pass is STFT o iSTFT, a
perfect-reconstruction pair for the Hann window at hop = n_fft/4 — after the first inverse transform the
signal only moves by rounding error.  The restatement keeps the loop exactly as written.

``librosa.istft`` / ``librosa.stft`` resolve inside ``librosa==0.10.2.post1`` (``requirements.txt:10``), not
installed here.  Published algorithm of ``istft`` restated below (defaults: ``n_fft = 2*(F-1)``,
``win_length = n_fft``, periodic Hann, ``center=True``, ``length=None``):

* every frame: ``irfft(S[:, f], n=n_fft)`` (imaginary parts of the DC and Nyquist bins are ignored, scale
  1/n_fft) times the float64 window, overlap-added at ``f*hop`` into a float32 signal of length
  ``n_fft + hop*(T-1)``;
* divided, where it exceeds ``tiny``, by ``window_sumsquare`` = sum of squared windows at the same offsets;
* ``n_fft//2`` samples trimmed from both ends -> length ``hop*(T-1)``.

STATUS: **parity unpinned** at the librosa boundary (no fixture in the reference covers it; the unseeded
``np.random.rand`` makes the reference's own output non-reproducible).  ``tests/test_stft_oracle.py`` cross-checks
``istft`` against ``torch.istft`` (which follows librosa's semantics) and the round trip against the identity.
"""
from __future__ import annotations

import numpy as np

from .stft_numpy import hann_periodic


def stft_complex(y: np.ndarray, n_fft: int, hop: int, center: bool = True) -> np.ndarray:
    """``y`` (L,) fp32 -> (1 + n_fft/2, n_frames) complex64 (librosa.stft)."""
    y = np.asarray(y, dtype=np.float32)
    if center:
        y = np.pad(y, n_fft // 2, mode="constant")
    if len(y) < n_fft:
        raise ValueError("audio shorter than n_fft")
    nfr = 1 + (len(y) - n_fft) // hop
    idx = np.arange(n_fft)[:, None] + hop * np.arange(nfr)[None, :]
    return np.fft.rfft(hann_periodic(n_fft)[:, None] * y[idx], axis=0).astype(np.complex64)


def istft_length(n_frames: int, hop: int) -> int:
    return hop * (n_frames - 1)


def istft(spec: np.ndarray, hop: int) -> np.ndarray:
    """``spec`` (F, T) complex -> (hop*(T-1),) float32 (librosa.istft, center=True, length=None)."""
    spec = np.asarray(spec)
    n_bins, n_frames = spec.shape
    n_fft = 2 * (n_bins - 1)
    win = hann_periodic(n_fft)
    y = np.zeros(n_fft + hop * (n_frames - 1), dtype=np.float32)
    wss = np.zeros_like(y)
    win_sq = (win ** 2)
    for f in range(n_frames):
        ytmp = win * np.fft.irfft(spec[:, f], n=n_fft)          # float64 * (float32|float64)
        y[f * hop:f * hop + n_fft] += ytmp                       # accumulated into the float32 signal
        wss[f * hop:f * hop + n_fft] += win_sq
    nz = wss > np.finfo(np.float32).tiny
    y[nz] /= wss[nz]
    pad = n_fft // 2
    return y[pad:len(y) - pad].copy()


def griffin_lim(magnitude: np.ndarray, n_fft: int, hop: int, iterations: int, rand: np.ndarray) -> np.ndarray:
    """``magnitude`` (F, T) fp32, ``rand`` (F, T) uniform [0,1) (what ``np.random.rand`` returned) -> audio fp32."""
    angles = np.exp(2j * np.pi * np.asarray(rand, dtype=np.float64))
    spec = magnitude * angles
    for _ in range(iterations):
        audio = istft(spec, hop)
        z = stft_complex(audio, n_fft, hop, True)
        spec = np.abs(z) * np.exp(1j * np.angle(z))
    return istft(spec, hop)
