"""Per-clip L1 on the device: the value each rank contributes to the multi-GPU all-gather.

``per_clip_l1(a, b)[i] = mean |a[i] - b[i]|``; clips have equal sizes, so the mean over clips equals the batch
``F.l1_loss`` term of the reference's ``CombinedPerceptualLoss`` (``/root/reference/code/loss.py:86``).
"""
from __future__ import annotations

import torch

from . import _lib


def per_clip_l1(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    if a.shape != b.shape or not a.is_cuda or not b.is_cuda or a.dtype != torch.float32 or b.dtype != torch.float32:
        raise ValueError("per_clip_l1: expected two same-shaped float32 tensors on a ROCm device")
    a = a.contiguous()
    b = b.contiguous()
    n = a.shape[0]
    elems = a[0].numel()
    out = torch.empty(n, dtype=torch.float32, device=a.device)
    stream = torch.cuda.current_stream(a.device).cuda_stream
    with torch.cuda.device(a.device):
        _lib.check(_lib.load().adn_per_clip_l1(a.data_ptr(), b.data_ptr(), n, elems, out.data_ptr(), stream),
                   "adn_per_clip_l1")
    return out
