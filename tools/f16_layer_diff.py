"""Per-layer comparison of the two fp16 3x3 kernel families (conv16_f16 vs conv_dma<_Float16>) on the same inputs and weights:
prints, per tap, the largest difference and where the elements that differ by more than 2 % of the tap's range sit (rows /
columns / channels) -- the tool that located the store-data hazard of profiles/NOTES.md (round 4).  Run on the GPU box:
    python tools/f16_layer_diff.py
"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from audiodenoiser_amd.model import UNet
from audiodenoiser_amd.weights import make_state_dict, make_input
dev = torch.device("cuda:0")
w = make_state_dict(0)
def mk(conv):
    os.environ["ADN_F16_CONV"] = conv
    m = UNet(1, 1)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in w.items()}, strict=True)
    m = m.to(dev).eval().set_compute_dtype("f16")
    with torch.no_grad(): m(torch.zeros((1, 1, 16, 16), device=dev))
    return m
a, b = mk("32"), mk("16")
for (n, f, t) in ((1, 64, 80), (1, 33, 47), (2, 32, 32)) * 2 + ((4, 257, 188), (2, 513, 256), (6, 257, 188), (3, 513, 256)) * 3:
    x = torch.from_numpy(make_input(7, n, f, t)).to(dev)
    with torch.no_grad():
        ya, ta = a(x, return_taps=True); yb, tb = b(x, return_taps=True)
    for k in ta:
        A, B = ta[k].float().cpu().numpy(), tb[k].float().cpu().numpy()
        d = np.abs(A - B)
        bad = np.argwhere(d > 0.02 * np.abs(A).max())
        if len(bad) or k == "out": print(f, t, k, A.shape, "maxdiff %.4g of %.4g" % (d.max(), np.abs(A).max()), "nbad", len(bad))
        if len(bad):
            print("   first bad idx", bad[:6].tolist(), "rows", sorted(set(bad[:, 2].tolist()))[:40], "cols", sorted(set(bad[:, 3].tolist()))[:40], "ch", sorted(set(bad[:, 1].tolist()))[:40])
