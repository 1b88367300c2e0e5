"""Griffin-Lim reconstruction with the reference's call surface (``/root/reference/code/test.py:29-48``), on HIP.

``griffin_lim_reconstruction(magnitude_spectrogram, n_fft, hop_length, iterations=50)`` takes the reference's
``(freq_bins, time_frames)`` magnitude (or a batch ``(N, F, T)``) and returns 1-D audio (``(N, L)`` for a
batch) of ``hop_length * (time_frames - 1)`` samples — numpy in -> numpy out, CUDA tensor in -> CUDA tensor out.
The loop is the reference's, including the fact that it never re-imposes the target magnitude (each pass is
``stft(istft(.))``); the random start phase is drawn like the reference (``np.random.rand``) unless ``rand`` is
given, which makes the result reproducible and testable.  ``istft`` / ``stft_complex`` expose the two transforms
(frame-major complex layout ``(n_clips, n_frames, n_bins)``).
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib

__all__ = ["griffin_lim_reconstruction", "istft", "stft_complex"]


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def griffin_lim_reconstruction(magnitude_spectrogram, n_fft, hop_length, iterations=50, rand=None, device=None):
    is_np = not isinstance(magnitude_spectrogram, torch.Tensor)
    mag = (torch.from_numpy(np.ascontiguousarray(magnitude_spectrogram, dtype=np.float32)).to(device or "cuda")
           if is_np else magnitude_spectrogram)
    if not mag.is_cuda:
        raise RuntimeError("griffin_lim_reconstruction: the magnitude must live on a ROCm device (no CPU path)")
    if mag.dtype != torch.float32:
        raise TypeError("griffin_lim_reconstruction: expected float32 magnitudes")
    single = mag.dim() == 2
    m = (mag[None] if single else mag).contiguous()
    if m.dim() != 3:
        raise ValueError("griffin_lim_reconstruction: magnitude must be (F, T) or (N, F, T)")
    n, f, t = m.shape
    if rand is None:
        rand = np.random.rand(n, f, t)                 # test.py:36, unseeded like the reference
    r = rand if isinstance(rand, torch.Tensor) else torch.from_numpy(np.asarray(rand, dtype=np.float32))
    r = r.to(device=m.device, dtype=torch.float32).reshape(n, f, t).contiguous()
    L = _lib.load()
    need, length = ctypes.c_size_t(), ctypes.c_long()
    _lib.check(L.adn_griffin_lim_workspace_bytes(n, f, t, ctypes.byref(need)), "adn_griffin_lim_workspace_bytes")
    _lib.check(L.adn_istft_length(t, hop_length, ctypes.byref(length)), "adn_istft_length")
    ws = torch.empty(need.value, dtype=torch.uint8, device=m.device)
    out = torch.empty((n, length.value), dtype=torch.float32, device=m.device)
    with torch.cuda.device(m.device):
        _lib.check(L.adn_griffin_lim(m.data_ptr(), r.data_ptr(), n, f, t, n_fft, hop_length, iterations, ws.data_ptr(),
                                     ws.numel(), out.data_ptr(), _stream(m.device)), "adn_griffin_lim")
    out = out[0] if single else out
    return out.cpu().numpy() if is_np else out


def stft_complex(audio: torch.Tensor, n_fft: int = 512, hop_length: int = 128) -> torch.Tensor:
    """``audio`` (n_clips, L) float32 on a ROCm device -> complex64 (n_clips, n_frames, n_fft/2+1), centred."""
    if not audio.is_cuda or audio.dtype != torch.float32 or audio.dim() != 2:
        raise ValueError("stft_complex: expected a (n_clips, L) float32 tensor on a ROCm device")
    a = audio.contiguous()
    n, length = a.shape
    t = 1 + length // hop_length
    out = torch.empty((n, t, n_fft // 2 + 1, 2), dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        _lib.check(_lib.load().adn_stft_complex(a.data_ptr(), n, length, n_fft, hop_length, out.data_ptr(),
                                                _stream(a.device)), "adn_stft_complex")
    return torch.view_as_complex(out)


def istft(spec: torch.Tensor, hop_length: int = 128) -> torch.Tensor:
    """``spec`` complex64 (n_clips, n_frames, n_bins) on a ROCm device -> (n_clips, hop*(n_frames-1)) float32."""
    if not spec.is_cuda or spec.dtype != torch.complex64 or spec.dim() != 3:
        raise ValueError("istft: expected a (n_clips, n_frames, n_bins) complex64 tensor on a ROCm device")
    s = torch.view_as_real(spec.contiguous())
    n, t, f = spec.shape
    n_fft = 2 * (f - 1)
    L = _lib.load()
    need = ctypes.c_size_t()
    _lib.check(L.adn_istft_workspace_bytes(n, t, n_fft, ctypes.byref(need)), "adn_istft_workspace_bytes")
    ws = torch.empty(need.value, dtype=torch.uint8, device=spec.device)
    out = torch.empty((n, hop_length * (t - 1)), dtype=torch.float32, device=spec.device)
    with torch.cuda.device(spec.device):
        _lib.check(L.adn_istft(s.data_ptr(), n, t, n_fft, hop_length, ws.data_ptr(), ws.numel(), out.data_ptr(),
                               _stream(spec.device)), "adn_istft")
    return out
