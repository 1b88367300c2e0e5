"""torch.nn.functional restatement of the reference forward — TEST INFRASTRUCTURE (see oracle/__init__.py).

Written from the behaviour of ``/root/reference/code/model.py:7-94`` as a flat function over a state
dict (no nn.Module tree), so it needs nothing from the reference at run time and can travel to the GPU
box.  On CPU each call dispatches to the same ATen/oneDNN kernels the reference's modules would, which
makes it the fair "reference CPU path" stand-in for ``bench.py``'s ``cpu_baseline`` (kind "port") and a
fast full-size checker.  tests/test_oracle_golden.py verifies it against outputs of the imported
reference frozen in tests/golden/.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from audiodenoiser_amd.weights import BN_EPS

_DOWN = ("downconv1", "downconv2", "downconv3", "downconv4")
_UP = ("upconv1", "upconv2", "upconv3", "upconv4")


def to_torch_state(sd, dtype=torch.float32, device="cpu"):
    out = {}
    for k, v in sd.items():
        t = v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))
        if t.is_floating_point():
            t = t.to(dtype)
        out[k] = t.to(device)
    return out


def _double_conv(sd, prefix, x):
    for conv, bn in ((0, 1), (3, 4)):
        p = f"{prefix}.double_conv."
        x = F.conv2d(x, sd[f"{p}{conv}.weight"], sd[f"{p}{conv}.bias"], stride=1, padding=1)
        x = F.batch_norm(x, sd[f"{p}{bn}.running_mean"], sd[f"{p}{bn}.running_var"],
                         sd[f"{p}{bn}.weight"], sd[f"{p}{bn}.bias"], training=False, eps=BN_EPS)
        x = F.relu(x)
    return x


@torch.no_grad()
def unet_forward(sd, x: torch.Tensor, want_taps: bool = False):
    """Eval-mode forward. ``sd``: torch state dict (``to_torch_state``); ``x`` (N,1,F,T)."""
    taps = {}
    skips = []
    cur = x
    for i, name in enumerate(_DOWN):
        s = _double_conv(sd, f"{name}.conv", cur)          # model.py:29-32
        skips.append(s)
        taps[f"down{i + 1}"] = s
        cur = F.max_pool2d(s, 2)
    cur = _double_conv(sd, "bottleneck", cur)              # model.py:81
    taps["bottleneck"] = cur
    for i, name in enumerate(_UP):
        x2 = skips[3 - i]
        x1 = F.conv_transpose2d(cur, sd[f"{name}.up.weight"], sd[f"{name}.up.bias"], stride=2)
        dy = x2.shape[2] - x1.shape[2]
        dx = x2.shape[3] - x1.shape[3]
        x1 = F.pad(x1, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])      # model.py:44-47
        cur = _double_conv(sd, f"{name}.conv", torch.cat([x2, x1], dim=1))  # skip channels first
        taps[f"up{i + 1}"] = cur
    y = F.conv2d(cur, sd["out.weight"], sd["out.bias"])    # model.py:93
    taps["out"] = y
    return (y, taps) if want_taps else y
