"""Shapes beyond the limits the library had through round 4 (T <= 4094, F*T < 2^24, in_channels * (T + 2) <= 4096): the
reference's network is fully convolutional (/root/reference/code/model.py:70-94) and test.py:112-113 feeds it whole clips, so any
F, T >= 16 must work.  Cheap shapes that cross each old limit are checked against the torch oracle; images too large for a CPU
forward are checked through a size-independent property of the domain: the network is translation equivariant away from the
borders, so a window of the big image, computed on its own, must reproduce the big result in its interior (beyond the 92-pixel
reach of the receptive field).  The window itself is checked against the oracle.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return torch.device("cuda", 0)


def _net(sd, dev, dtype="f32", cin=1, k=1):
    from audiodenoiser_amd.model import UNet
    m = UNet(cin, k)
    m.load_state_dict({kk: torch.from_numpy(np.array(v)) for kk, v in sd.items()}, strict=True)
    return m.to(dev).eval().set_compute_dtype(dtype)


def _rel(a, ref):
    return float(np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-30))


@pytest.mark.parametrize("dtype,tol", [("f32", 1e-4), ("f16", 1e-2)])
@pytest.mark.parametrize("f,t", [(16, 8192), (32, 5000), (20, 4095), (17, 65535)])
def test_long_clips_match_oracle(dev, weights_np, f, t, dtype, tol):
    """T beyond 4094 (the first layer's input window no longer has to hold whole rows: column tiles)."""
    from oracle import unet_torch
    from audiodenoiser_amd.weights import make_input
    x = make_input(11, 1, f, t)
    ref = unet_torch.unet_forward(unet_torch.to_torch_state(weights_np), torch.from_numpy(x)).numpy()
    m = _net(weights_np, dev, dtype)
    with torch.no_grad():
        y = m(torch.from_numpy(x).to(dev)).cpu().numpy()
    assert _rel(y, ref) <= tol


@pytest.mark.parametrize("dtype,tol", [("f32", 1e-4), ("f16", 1e-2)])
def test_many_input_planes_on_a_long_clip(dev, dtype, tol):
    """UNet(in_channels=3, num_classes=2) at T = 4500: in_channels * (T + 2) > 4096, the old bound of the first layer."""
    from oracle import unet_torch
    from audiodenoiser_amd.weights import make_input, make_state_dict
    sd = make_state_dict(1234, 3, 2)
    x = make_input(12, 3, 24, 4500).reshape(1, 3, 24, 4500)
    ref = unet_torch.unet_forward(unet_torch.to_torch_state(sd), torch.from_numpy(x)).numpy()
    m = _net(sd, dev, dtype, 3, 2)
    with torch.no_grad():
        y = m(torch.from_numpy(x).to(dev)).cpu().numpy()
    assert y.shape == ref.shape and _rel(y, ref) <= tol


def test_twenty_input_planes(dev):
    """UNet(in_channels=20, num_classes=2): beyond the 15 planes the first layer's input window was limited to through round 4."""
    from oracle import unet_torch
    from audiodenoiser_amd.weights import make_input, make_state_dict
    sd = make_state_dict(1234, 20, 2)
    x = make_input(13, 20, 40, 56).reshape(1, 20, 40, 56)
    ref = unet_torch.unet_forward(unet_torch.to_torch_state(sd), torch.from_numpy(x)).numpy()
    for dtype, tol in (("f32", 1e-4), ("f16", 1e-2)):
        with torch.no_grad():
            y = _net(sd, dev, dtype, 20, 2)(torch.from_numpy(x).to(dev)).cpu().numpy()
        assert y.shape == ref.shape and _rel(y, ref) <= tol, dtype


# (F, T), dtype: 4100^2 > 2^24 pixels (the old bound: a 64-channel fp32 image of 4.3 GB, beyond one buffer descriptor);
# 6704^2 = 44.9 M pixels: beyond conv16_f16's 32-bit output offsets (the fp16 path falls back to conv_dma<_Float16> and the unfused
# first layer) and a 1.4 GB channel block; 11584^2 = 134.19 M pixels: just below 2^27, channel blocks of 4.29 GB -- byte offsets up
# to 2^32 - 2^20; 1 100 000 x 16: a 64-channel image of 4.5 GB only 16 pixels wide -- too large for wino4_conv_f32's pair mode (two
# clips through one descriptor), also with F(4x4,3x3) forced on every layer ("f32:f4").  Workspace: 1.1 KB per pixel in fp32 (149 GB
# for the largest), half of it in fp16.
BIG = [((4100, 4100), "f32"), ((4100, 4100), "f16"), ((6704, 6704), "f32"), ((6704, 6704), "f16"), ((11584, 11584), "f16"),
       ((11584, 11584), "f32"), ((48, 2796000), "f32"), ((1100000, 16), "f32"), ((1100000, 16), "f32:f4")]
WIN, MARGIN = 608, 96


@pytest.mark.parametrize("shape,dtype", BIG, ids=[f"{s[0]}x{s[1]}-{d.replace(':', '-')}" for s, d in BIG])
def test_images_beyond_one_buffer_descriptor(dev, weights_np, shape, dtype, monkeypatch):
    from oracle import unet_torch
    f, t = shape
    monkeypatch.delenv("ADN_WINO_TILE", raising=False)
    if dtype.endswith(":f4"):
        dtype = dtype.split(":")[0]
        monkeypatch.setenv("ADN_WINO_TILE", "4")
    tol = 1e-4 if dtype == "f32" else 1e-2
    need = f * t * (1112 if dtype == "f32" else 560) + 3 * f * t * 4
    free = torch.cuda.mem_get_info(dev)[0]
    if need > free * 0.95:
        pytest.skip(f"{f}x{t} {dtype} needs {need / 2**30:.0f} GiB of device memory, {free / 2**30:.0f} free")
    m = _net(weights_np, dev, dtype)
    g = torch.Generator(device=dev).manual_seed(f * 31 + t)
    x = torch.rand((1, 1, f, t), generator=g, device=dev) * 3
    with torch.no_grad():
        y = m(x)
        assert bool(torch.isfinite(y).all())
        wh, ww = min(WIN, f), min(WIN, t)
        # windows at the far corner (largest addresses), in the middle and at the origin; origins on the 16-pixel grid of the four poolings
        corners = {((f - wh) // 16 * 16, (t - ww) // 16 * 16), ((f // 2) // 16 * 16 if f > wh else 0, (t // 2) // 16 * 16 if t > ww else 0), (0, 0)}
        worst = 0.0
        for r0, c0 in sorted(corners):
            r0, c0 = min(r0, f - wh), min(c0, t - ww)
            xw = x[:, :, r0:r0 + wh, c0:c0 + ww].contiguous()
            yw = m(xw)
            # interior of the window: MARGIN pixels from every window edge that is not also an edge of the big image
            a0 = 0 if r0 == 0 else MARGIN
            a1 = wh if r0 + wh == f else wh - MARGIN
            b0 = 0 if c0 == 0 else MARGIN
            b1 = ww if c0 + ww == t else ww - MARGIN
            assert a1 > a0 and b1 > b0
            big = y[0, 0, r0 + a0:r0 + a1, c0 + b0:c0 + b1]
            small = yw[0, 0, a0:a1, b0:b1]
            e = float((big - small).abs().max()) / float(small.abs().max())
            worst = max(worst, e)
            assert e <= tol, (shape, dtype, (r0, c0), e)
            if (r0, c0) == sorted(corners)[-1]:               # ... and that window against the CPU oracle
                ref = unet_torch.unet_forward(unet_torch.to_torch_state(weights_np), xw.cpu()).numpy()
                assert _rel(yw.cpu().numpy(), ref) <= tol
    print(f"big image {f}x{t} {dtype}: window / whole-image agreement {worst:.2e} of max|y| (bound {tol:g})")
    del y, x
    m._workspace = None
    torch.cuda.empty_cache()


AGREE_SHAPES = [(1, 16, 16), (1, 33, 47), (2, 64, 80), (1, 100, 300), (3, 129, 65), (1, 257, 188), (1, 513, 256), (5, 257, 188),
                (16, 256, 64), (1, 1025, 16), (7, 48, 1040), (1, 40, 2000), (2, 513, 256), (4, 300, 200)]


@pytest.mark.parametrize("dtype,tol", [("f32", 2e-5), ("f16", 5e-3)])
def test_automatic_and_pinned_kernel_choice_agree(dev, weights_np, dtype, tol):
    """The automatic choice (small grids: finer tiles, K loops cut over several workgroups in four kernel families + reduce launches)
    against one kernel per layer (set_batch_invariant): the same network to the last bits of fp32 (2e-5 of max|y|) / of fp16 storage
    (5e-3), on shapes from one tile to the reference's own batches (5 x 257x188, 16 x 256x64) -- every slice bound, concat switch
    and partial-buffer layout is crossed by some shape here."""
    from audiodenoiser_amd.weights import make_input
    auto = _net(weights_np, dev, dtype)
    pinned = _net(weights_np, dev, dtype).set_batch_invariant(True)
    worst = 0.0
    for n, f, t in AGREE_SHAPES:
        x = torch.from_numpy(make_input(300 + f, n, f, t)).to(dev)
        with torch.no_grad():
            ya, yp = auto(x), pinned(x)
        assert bool(torch.isfinite(ya).all()) and ya.shape == yp.shape
        err = float((ya - yp).abs().max() / yp.abs().max().clamp_min(1e-30))
        worst = max(worst, err)
        assert err <= tol, (dtype, n, f, t, err)
    print(f"automatic vs pinned kernel choice, {dtype}: worst {worst:.2e} of max|y| (bound {tol})")
