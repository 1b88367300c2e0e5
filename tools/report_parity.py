#!/usr/bin/env python3
"""Print the measured parity of the HIP path against the committed reference goldens (run on the GPU box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from audiodenoiser_amd.model import UNet  # noqa: E402
from audiodenoiser_amd.weights import make_input, make_state_dict  # noqa: E402

sd = make_state_dict(1234)
net = UNet()
net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
net = net.cuda().eval()
algo = os.environ.get("ADN_CONV_ALGO", "winograd")
for (n, f, t) in ((2, 16, 16), (2, 33, 47), (1, 64, 80), (1, 257, 188), (1, 513, 256)):
    g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", f"unet_{f}x{t}.npz"))
    with torch.no_grad():
        y = net(torch.from_numpy(make_input(7, n, f, t)).cuda()).cpu().numpy()
    print(f"{algo:9s} {n}x1x{f}x{t}: max|y-ref|/max|ref| = {np.abs(y - g['y']).max() / np.abs(g['y']).max():.2e}")

# fp16 path (BASELINE configs[4], tolerance 1e-2) and the real-audio configs[0] golden (reference data_loader + model)
net16 = UNet()
net16.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
net16 = net16.cuda().eval().set_compute_dtype("f16")
for (n, f, t) in ((1, 257, 188), (1, 513, 256)):
    g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", f"unet_{f}x{t}.npz"))
    with torch.no_grad():
        y = net16(torch.from_numpy(make_input(7, n, f, t)).cuda()).cpu().numpy()
    print(f"{'fp16 path':9s} {n}x1x{f}x{t}: max|y-ref|/max|ref| = {np.abs(y - g['y']).max() / np.abs(g['y']).max():.2e}")
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "config0_real_audio.npz"))
with torch.no_grad():
    y = net(torch.from_numpy(g["x_f16"].astype(np.float32)[None, None]).cuda()).cpu().numpy()[0, 0]
print(f"{algo:9s} config0 real audio 513x256: max|y-ref|/max|ref| = {np.abs(y - g['y']).max() / np.abs(g['y']).max():.2e}")

# goldens under trained-like BatchNorm statistics / heavy-tailed weights, real-audio input at scale 1 and 100 (per clip)
from audiodenoiser_amd.weights import make_state_dict_variant  # noqa: E402
gd = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")
x16 = np.load(os.path.join(gd, "config0_real_audio.npz"))["x_f16"]
for kind in ("trained", "heavy"):
    sdv = make_state_dict_variant(kind, 1234)
    nv = UNet()
    nv.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sdv.items()})
    nv = nv.cuda().eval()
    for (f, t) in ((33, 47), (257, 188), (513, 256)):
        g = np.load(os.path.join(gd, f"unet_{kind}_{f}x{t}.npz"))
        x = x16[:f, :t].astype(np.float32)
        xb = torch.from_numpy(np.stack([x, x * np.float32(100.0)])[:, None]).cuda()
        with torch.no_grad():
            y = nv(xb).cpu().numpy()
        e = [np.abs(y[c] - g["y"][c]).max() / np.abs(g["y"][c]).max() for c in range(2)]
        print(f"{algo:9s} weights '{kind}' 2x1x{f}x{t}: max|y-ref|/max|ref| = {e[0]:.2e} (x1)  {e[1]:.2e} (x100)")
