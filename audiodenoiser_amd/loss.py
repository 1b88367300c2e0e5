"""Per-clip losses on the device: the values each rank contributes to the multi-GPU all-gather and
the mirror of the reference's ``loss.CombinedPerceptualLoss`` (``/root/reference/code/loss.py:6-95``).

``per_clip_l1(a, b)[i] = mean |a[i] - b[i]|``; clips have equal sizes, so the mean over clips equals the batch
``F.l1_loss`` term of the reference's ``CombinedPerceptualLoss`` (``loss.py:86``).

CPU tensors (the reference's ``test.py:118-122`` builds its loss inputs on the CPU) are staged onto the current ROCm
device, computed there and the result is returned on the inputs' device; without a device every call raises.
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


def perceptual_loss_per_clip(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """(B,1,F,T) x2 on a ROCm device -> (B,4) = [total, stft, mel, l1] per clip (reference ``loss.py:6-95``)."""
    import ctypes
    if (pred.shape != target.shape or pred.dim() != 4 or pred.shape[1] != 1 or pred.device != target.device
            or pred.dtype != torch.float32 or target.dtype != torch.float32):
        raise ValueError("perceptual_loss_per_clip: expected two (B,1,F,T) float32 tensors on one device")
    home = pred.device
    if not pred.is_cuda:                                   # test.py:118-122: CPU tensors -> staged, HIP path, back
        dev = _lib.staging_device()
        return perceptual_loss_per_clip(pred.to(dev), target.to(dev)).to(home)
    pred = pred.contiguous()
    target = target.contiguous()
    b, _, f, t = pred.shape
    L = _lib.load()
    need = ctypes.c_size_t()
    _lib.check(L.adn_perceptual_loss_workspace_bytes(b, f, t, ctypes.byref(need)), "adn_perceptual_loss_workspace_bytes")
    ws = torch.empty(need.value, dtype=torch.uint8, device=pred.device)
    out = torch.empty((b, 4), dtype=torch.float32, device=pred.device)
    stream = torch.cuda.current_stream(pred.device).cuda_stream
    with torch.cuda.device(pred.device):
        _lib.check(L.adn_perceptual_loss(pred.data_ptr(), target.data_ptr(), b, f, t, ws.data_ptr(), ws.numel(),
                                         out.data_ptr(), stream), "adn_perceptual_loss")
    return out


class CombinedPerceptualLoss(torch.nn.Module):
    """Drop-in for the reference's ``loss.CombinedPerceptualLoss`` (``loss.py:71-95``) for evaluation:
    ``forward(pred, target) -> (total, stft, mel, l1)`` batch scalars, computed per clip on the device and averaged
    (identical to the reference's batch ``l1_loss`` values because clips have equal sizes).  No autograd."""

    def forward(self, pred, target):
        m = perceptual_loss_per_clip(pred, target).mean(dim=0)
        return m[0], m[1], m[2], m[3]
