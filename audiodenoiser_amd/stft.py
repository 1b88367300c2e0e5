"""STFT-magnitude helpers with the reference's call surface, computed by the fused HIP kernel.

Mirrors ``audio_to_magnitude_spectrogram`` (``/root/reference/code/create_train_dataset.py:162-174``,
``librosa.stft(..., center=False)`` + ``magphase``) and ``audio_to_spectrogram``
(``/root/reference/code/create_test_dataset.py:35-41``, ``center=True`` with librosa 0.10's zero padding).
Defaults are the reference's constants N_FFT=512, HOP=128 (``create_train_dataset.py:26-27``).
numpy in -> numpy out (like the reference); CUDA tensor in -> CUDA tensor out (no host round trip).
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib

N_FFT = 512
HOP_LENGTH = 128


def prepare(device=None, n_fft: int = N_FFT) -> None:
    """Build the constant tables of ``n_fft`` (and of the per-clip loss) on ``device`` ahead of time (``adn_prepare``): afterwards
    the STFT-family calls and the loss only enqueue on the current stream, so they can be recorded into a HIP graph."""
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    _lib.check(_lib.load().adn_prepare(dev.index if dev.index is not None else torch.cuda.current_device(), int(n_fft)),
               "adn_prepare")


def stft_n_frames(length: int, n_fft: int, hop: int, center: bool) -> int:
    out = ctypes.c_long()
    _lib.check(_lib.load().adn_stft_n_frames(length, n_fft, hop, 1 if center else 0, ctypes.byref(out)),
               "adn_stft_n_frames")
    return int(out.value)


def stft_magnitude(audio: torch.Tensor, n_fft: int = N_FFT, hop_length: int = HOP_LENGTH,
                   center: bool = True) -> torch.Tensor:
    """``audio`` (L,) or (n_clips, L) float32 on a ROCm device -> (..., n_fft/2+1, n_frames) float32."""
    if not audio.is_cuda:
        raise RuntimeError("stft_magnitude: audio must live on a ROCm device (no CPU path)")
    if audio.dtype != torch.float32:
        raise TypeError("stft_magnitude: expected float32 audio")
    single = audio.dim() == 1
    a = (audio[None] if single else audio).contiguous()
    if a.dim() != 2:
        raise ValueError("stft_magnitude: audio must be (L,) or (n_clips, L)")
    n_clips, length = a.shape
    nfr = stft_n_frames(length, n_fft, hop_length, center)
    if nfr <= 0:
        raise ValueError(f"audio of {length} samples is shorter than n_fft={n_fft}")
    out = torch.empty((n_clips, n_fft // 2 + 1, nfr), dtype=torch.float32, device=a.device)
    stream = torch.cuda.current_stream(a.device).cuda_stream
    with torch.cuda.device(a.device):
        _lib.check(_lib.load().adn_stft_mag(a.data_ptr(), n_clips, length, n_fft, hop_length, 1 if center else 0,
                                            out.data_ptr(), stream), "adn_stft_mag")
    return out[0] if single else out


def stft_magnitude_fit(audio: torch.Tensor, target_size, n_fft: int = N_FFT, hop_length: int = HOP_LENGTH,
                       center: bool = True) -> torch.Tensor:
    """``audio`` (n_clips, L) float32 on a ROCm device -> (n_clips, 1, H, W): the STFT magnitude with the loader rule of
    ``SpectrogramDataset`` (fp16 round trip, crop / bottom-right zero pad to ``target_size``; reference
    ``data_loader.py:41-42,54-72``) applied in the same kernel.  Equal, bit for bit, to
    ``quantize_pad_on_device(stft_magnitude(audio, ...), target_size)`` but computes only the frames inside the window."""
    if not audio.is_cuda or audio.dtype != torch.float32 or audio.dim() != 2:
        raise ValueError("stft_magnitude_fit: expected a (n_clips, L) float32 tensor on a ROCm device")
    a = audio.contiguous()
    n_clips, length = a.shape
    H, W = (int(v) for v in target_size)
    if stft_n_frames(length, n_fft, hop_length, center) <= 0:
        raise ValueError(f"audio of {length} samples is shorter than n_fft={n_fft}")
    out = torch.empty((n_clips, 1, H, W), dtype=torch.float32, device=a.device)
    stream = torch.cuda.current_stream(a.device).cuda_stream
    with torch.cuda.device(a.device):
        _lib.check(_lib.load().adn_stft_mag_fit(a.data_ptr(), n_clips, length, n_fft, hop_length, 1 if center else 0,
                                                out.data_ptr(), H, W, stream), "adn_stft_mag_fit")
    return out


def _dispatch(audio, n_fft, hop_length, center, device):
    if isinstance(audio, torch.Tensor):
        return stft_magnitude(audio, n_fft, hop_length, center)
    a = torch.from_numpy(np.ascontiguousarray(audio, dtype=np.float32)).to(device or "cuda")
    return stft_magnitude(a, n_fft, hop_length, center).cpu().numpy()


def audio_to_magnitude_spectrogram(audio_1d, n_fft: int = N_FFT, hop_length: int = HOP_LENGTH, device=None):
    """1-D audio -> (n_fft/2+1, 1 + (L - n_fft)//hop) magnitudes, no centring (train-set builder semantics)."""
    return _dispatch(audio_1d, n_fft, hop_length, False, device)


def audio_to_spectrogram(audio, n_fft: int = N_FFT, hop_length: int = HOP_LENGTH, device=None):
    """1-D audio -> (n_fft/2+1, 1 + L//hop) magnitudes, centred with zero padding (test-set builder semantics)."""
    return _dispatch(audio, n_fft, hop_length, True, device)
