#!/usr/bin/env python3
"""Per-launch times of the forward at small batches under the three kernel-choice modes (GPU box):
default (automatic small-grid rule, ADN_AUTO_GRID workgroups), ADN_BATCH_INVARIANT=1, ADN_WINO_TILE=2 + ADN_WINO_SPLITK=1.
Calibrates the threshold of choose_algo (csrc/adn_api.hip).  -> stdout
    python tools/small_grid_probe.py [auto_grid ...]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from audiodenoiser_amd import _lib  # noqa: E402
from audiodenoiser_amd.model import UNet  # noqa: E402
from audiodenoiser_amd.roofline import unet_launches  # noqa: E402
from audiodenoiser_amd.weights import make_state_dict  # noqa: E402

SHAPES = ((1, 513, 256), (2, 513, 256), (4, 513, 256), (8, 513, 256), (16, 513, 256), (1, 257, 188), (5, 257, 188), (16, 256, 64), (64, 256, 64))


def run(env, sd, dev, iters=20):
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        net = UNet(1, 1)
        net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, strict=True)
        net = net.to(dev).eval()
        L = _lib.load()
        out = {}
        with torch.no_grad():
            for b, f, t in SHAPES:
                x = torch.rand((b, 1, f, t), device=dev) * 4
                for _ in range(3):
                    net(x)
                _lib.check(L.adn_unet_set_timing(net._handle, iters), "set_timing")
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(iters):
                    net(x)
                e1.record()
                torch.cuda.synchronize()
                ms = np.zeros((iters, 23), dtype=np.float32)
                for i in range(iters):
                    _lib.check(L.adn_unet_get_timing(net._handle, i, ms[i].ctypes.data_as(_lib.c_float_p)), "get_timing")
                _lib.check(L.adn_unet_set_timing(net._handle, 0), "set_timing")
                out[(b, f, t)] = (e0.elapsed_time(e1) / iters, ms.mean(axis=0))
        net._release()
        return out
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def main():
    dev = torch.device("cuda", 0)
    sd = make_state_dict(1234)
    modes = [("invariant", {"ADN_BATCH_INVARIANT": "1"}), ("serving", {"ADN_WINO_TILE": "2", "ADN_WINO_SPLITK": "1"})]
    for g in (sys.argv[1:] or ["512"]):
        modes.append((f"auto{g}", {"ADN_AUTO_GRID": g}))
    res = {name: run(env, sd, dev) for name, env in modes}
    names = [l["name"] for l in unet_launches(513, 256)]
    for shp in SHAPES:
        print(f"== batch {shp[0]} x {shp[1]}x{shp[2]}: ms per forward  " + "  ".join(f"{m} {res[m][shp][0]:.3f}" for m, _ in modes))
        for i, nm in enumerate(names):
            print(f"   {nm:22s} " + "  ".join(f"{res[m][shp][1][i]:8.4f}" for m, _ in modes))


if __name__ == "__main__":
    main()
