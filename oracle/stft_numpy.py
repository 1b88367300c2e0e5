"""numpy restatement of ``librosa.stft`` + ``librosa.magphase`` — TEST INFRASTRUCTURE (oracle/__init__.py).

The reference calls (``/root/reference/code/create_train_dataset.py:167-173`` with ``center=False`` and
``/root/reference/code/create_test_dataset.py:39-40`` with the default ``center=True``) resolve inside
``librosa==0.10.2.post1`` (``requirements.txt:10``), which is not installed here and cannot be fetched.
This restates its published algorithm:

* ``window = scipy.signal.get_window("hann", n_fft, fftbins=True)`` — periodic Hann, float64,
  ``win_length = n_fft`` so no centre padding of the window;
* ``center=True`` -> ``np.pad(y, n_fft // 2, mode="constant")`` (0.10 default ``pad_mode="constant"``);
* frames of ``n_fft`` samples every ``hop`` samples: ``n_frames = 1 + (len - n_fft) // hop``;
* ``np.fft.rfft(window * frames, axis=0)`` — the float64 window promotes the product to float64;
* the result is stored into a ``complex64`` matrix (``util.dtype_r2c(float32)``) of shape
  ``(1 + n_fft/2, n_frames)``;
* ``magphase`` -> ``np.abs(D)`` (power 1) -> float32.

STATUS: **parity unpinned** at the librosa boundary — no fixture of the reference covers it.
"""
from __future__ import annotations

import numpy as np


def hann_periodic(n_fft: int) -> np.ndarray:
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n_fft, dtype=np.float64) / n_fft)


def n_frames(length: int, n_fft: int, hop: int, center: bool) -> int:
    lp = length + 2 * (n_fft // 2) if center else length
    return 0 if lp < n_fft else 1 + (lp - n_fft) // hop


def stft_mag(y: np.ndarray, n_fft: int, hop: int, center: bool) -> np.ndarray:
    """``y`` (L,) fp32 -> (1 + n_fft/2, n_frames) fp32 magnitude."""
    y = np.asarray(y, dtype=np.float32)
    if center:
        y = np.pad(y, n_fft // 2, mode="constant")
    nfr = 1 + (len(y) - n_fft) // hop
    if len(y) < n_fft:
        raise ValueError("audio shorter than n_fft")
    idx = np.arange(n_fft)[:, None] + hop * np.arange(nfr)[None, :]
    frames = y[idx]                                            # (n_fft, n_frames) float32
    spec = np.fft.rfft(hann_periodic(n_fft)[:, None] * frames, axis=0)   # float64 -> complex128
    return np.abs(spec.astype(np.complex64)).astype(np.float32)
