"""Drop-in for the reference's ``data_loader.py`` (``/root/reference/code/data_loader.py:1-72``).

``SpectrogramDataset(data_dir, target_size=(256, 64))`` keeps the reference contract: it pairs the sorted
``clean*.npy`` / ``noisy*.npy`` files of one folder, and ``__getitem__`` returns ``(noisy, clean)`` as
``(1, H, W)`` float32 tensors whose values went through float16 and were cropped / zero padded (bottom, right)
to ``target_size``.  That per-item host path is file I/O plus two numpy calls and stays on the host, as in the
reference (DataLoader worker processes cannot share a GPU context).

``load_batch_to_device`` is the MI355X ingest for the step right before the forward: raw fp32 spectrograms
are uploaded once and quantised + cropped/padded by one HIP kernel (``adn_quantize_pad``), producing the
``(B, 1, H, W)`` batch directly in HBM.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np
import torch
from torch.utils.data import Dataset

from . import _lib


def _list(data_dir: str, prefix: str):
    return sorted(os.path.join(data_dir, f) for f in os.listdir(data_dir)
                  if f.startswith(prefix) and f.endswith(".npy"))


def fit_to(data: np.ndarray, target_size) -> np.ndarray:
    """Crop or zero-pad (bottom / right) a 2-D array to ``target_size`` (reference ``_pad_or_truncate``)."""
    th, tw = target_size
    out = np.zeros((th, tw), dtype=data.dtype)
    h, w = min(th, data.shape[0]), min(tw, data.shape[1])
    out[:h, :w] = data[:h, :w]
    return out


class SpectrogramDataset(Dataset):
    def __init__(self, data_dir, target_size=(256, 64)):
        self.target_size = tuple(target_size)
        clean = _list(data_dir, "clean")
        noisy = _list(data_dir, "noisy")
        print(f"Found {len(clean)} clean files and {len(noisy)} noisy files in {data_dir}")
        assert len(clean) == len(noisy), f"Mismatch in {data_dir}"
        self.pairs = list(zip(noisy, clean))
        print(f"Total pairs loaded: {len(self.pairs)}")

    def __len__(self):
        return len(self.pairs)

    def _load(self, path):
        with np.errstate(over="ignore"):
            spec = np.load(path).astype(np.float16)       # honours the header's fortran_order flag
        return torch.from_numpy(fit_to(spec, self.target_size).astype(np.float32)).unsqueeze(0)

    def __getitem__(self, idx):
        noisy_path, clean_path = self.pairs[idx]
        return self._load(noisy_path), self._load(clean_path)

    # ---- MI355X ingest -------------------------------------------------------------------------------
    def load_batch_to_device(self, indices, device="cuda"):
        """(noisy, clean) batches ``(B, 1, H, W)`` float32 on ``device`` for same-shaped source files."""
        noisy = np.stack([np.ascontiguousarray(np.load(self.pairs[i][0]), dtype=np.float32) for i in indices])
        clean = np.stack([np.ascontiguousarray(np.load(self.pairs[i][1]), dtype=np.float32) for i in indices])
        return (quantize_pad_on_device(torch.from_numpy(noisy).to(device), self.target_size),
                quantize_pad_on_device(torch.from_numpy(clean).to(device), self.target_size))


def quantize_pad_on_device(spec: torch.Tensor, target_size) -> torch.Tensor:
    """``spec`` (B, h, w) float32 on a ROCm device -> (B, 1, H, W) = fp32(fp16(spec)) cropped / zero padded."""
    if not spec.is_cuda or spec.dtype != torch.float32 or spec.dim() != 3:
        raise ValueError("quantize_pad_on_device: expected a (B, h, w) float32 tensor on a ROCm device")
    spec = spec.contiguous()
    b, h, w = spec.shape
    H, W = target_size
    out = torch.empty((b, 1, H, W), dtype=torch.float32, device=spec.device)
    stream = torch.cuda.current_stream(spec.device).cuda_stream
    with torch.cuda.device(spec.device):
        _lib.check(_lib.load().adn_quantize_pad(spec.data_ptr(), b, h, w, out.data_ptr(), H, W, stream),
                   "adn_quantize_pad")
    return out
