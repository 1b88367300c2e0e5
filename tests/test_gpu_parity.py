"""Parity of the HIP path (through the C ABI) against the oracle and the reference-generated goldens.

Tolerances (BASELINE.json north_star: "within 1e-4 relative fp32"):
  U-Net  : max|y - ref| <= 1e-4 * max|ref| on the output; block outputs within 1e-4 of their rms scale.
  STFT   : max|m - ref| <= 1e-4 * max|ref| (oracle = float64 FFT rounded to fp32, librosa semantics; UNPINNED
           at the librosa boundary, see oracle/stft_numpy.py).
  loader : bit exact.
Nothing here reads /root/reference: goldens are committed under tests/golden/.
"""
import ctypes
import os
import sys

import numpy as np
import pytest

from conftest import load_real_audio_fixture
import torch

pytestmark = pytest.mark.gpu

TOL = 1e-4
GOLDEN_SHAPES = ((2, 16, 16), (2, 33, 47), (1, 64, 80), (1, 257, 188), (1, 513, 256))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def net(weights_np, dev):
    from audiodenoiser_amd.model import UNet
    m = UNet(1, 1)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in weights_np.items()}, strict=True)
    return m.to(dev).eval()


@pytest.fixture(scope="module")
def net_invariant(weights_np, dev):
    """UNet.set_batch_invariant() (adn_unet_set_batch_invariant): one kernel per layer by geometry alone, so a clip's result is
    bit-identical whatever batch it is computed in.  (The default picks finer-grained kernels for small grids.)"""
    from audiodenoiser_amd.model import UNet
    m = UNet(1, 1)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in weights_np.items()}, strict=True)
    return m.to(dev).eval().set_batch_invariant(True)


def _rel(a, ref):
    return float(np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-30))


@pytest.mark.parametrize("n,f,t", GOLDEN_SHAPES)
def test_unet_matches_reference_golden(net, dev, golden_dir, n, f, t):
    from audiodenoiser_amd.weights import make_input
    g = np.load(os.path.join(golden_dir, f"unet_{f}x{t}.npz"))
    x = torch.from_numpy(make_input(7, n, f, t)).to(dev)
    with torch.no_grad():
        y, taps = net(x, return_taps=True)
        y_plain = net(x)            # production path: the last two layers run fused (no up4 tensor is written)
    y = y.cpu().numpy()
    assert y.shape == g["y"].shape
    assert _rel(y, g["y"]) <= TOL
    assert _rel(y_plain.cpu().numpy(), g["y"]) <= TOL
    assert _rel(y_plain.cpu().numpy(), y) <= 1e-5          # fused vs unfused tail: summation order only
    for name, tp in taps.items():
        a = tp.cpu().numpy().astype(np.float64).ravel()
        s, sa, sq, cnt = g[f"{name}_stats"]
        assert a.size == int(cnt), name
        rms = np.sqrt(sq / cnt)
        assert np.abs(a[g[f"{name}_idx"]] - g[f"{name}_val"]).max() <= 10 * TOL * rms, name
        assert abs(np.abs(a).sum() - sa) <= TOL * sa, name
        assert abs((a * a).sum() - sq) <= 2 * TOL * sq, name


@pytest.mark.parametrize("n,f,t", [(3, 20, 36), (1, 48, 100), (2, 31, 16), (1, 16, 130)])
def test_unet_matches_oracle_all_blocks(net, dev, weights_np, n, f, t):
    """Seeded shapes not among the goldens (odd sizes, width-only / height-only pads), every block output."""
    import oracle
    from audiodenoiser_amd.weights import make_input
    x = make_input(21, n, f, t)
    ref, rtaps = oracle.unet_forward(weights_np, x, acc64=True, want_taps=True)
    with torch.no_grad():
        y, taps = net(torch.from_numpy(x).to(dev), return_taps=True)
        y_plain = net(torch.from_numpy(x).to(dev))      # production path: first and last layers fused into their neighbours
    for name in oracle.TAP_NAMES:
        a = taps[name].cpu().numpy()
        assert a.shape == rtaps[name].shape, name
        assert _rel(a, rtaps[name]) <= TOL, name
    assert _rel(y.cpu().numpy(), ref) <= TOL
    assert _rel(y_plain.cpu().numpy(), ref) <= TOL


def test_unet_full_size_batch64(net, net_invariant, dev, weights_np):
    """BASELINE config 2 (batch 64 x 513 x 256): size-independent properties + two clips against the oracle."""
    from oracle import unet_torch
    from audiodenoiser_amd.weights import make_input
    n, f, t = 64, 513, 256
    x = torch.from_numpy(make_input(0, n, f, t, scale=4.0)).to(dev)
    with torch.no_grad():
        y = net(x)
        yinv = net_invariant(x)
        assert torch.equal(y, yinv)                      # at this size the default handle runs the same kernels
        # (1) clip independence: with ADN_BATCH_INVARIANT=1 a clip computed alone is bit-identical to the same clip inside the
        # batch; the default handle computes a single clip with its small-grid kernels (F(2x2,3x3) + split-K): same tolerance
        # against the reference, last-bit differences against the batch
        for i in (0, 37, 63):
            assert torch.equal(net_invariant(x[i:i + 1].clone())[0], y[i]), i
            yi = net(x[i:i + 1].clone())
            assert float((yi[0] - y[i]).abs().max()) <= 2e-5 * float(y[i].abs().max()), i
        # the option is a property of the handle that may change between forwards (adn_unet_set_batch_invariant)
        net.set_batch_invariant(True)
        assert torch.equal(net(x[37:38].clone())[0], y[37])
        net.set_batch_invariant(False)
        y37 = net(x[37:38].clone())                      # back on the small-grid kernels: last-bit differences again, deterministic
        assert torch.equal(y37, net(x[37:38].clone())) and float((y37[0] - y[37]).abs().max()) <= 2e-5 * float(y[37].abs().max())
        # (2) batch order equivariance, bit exact
        perm = torch.randperm(n, generator=torch.Generator().manual_seed(1)).to(dev)
        assert torch.equal(net(x[perm].contiguous()), y[perm])
    assert torch.isfinite(y).all()
    # (3) oracle (same ATen/oneDNN kernels as the reference) on two clips
    sd = unet_torch.to_torch_state(weights_np)
    for i in (5, 63):
        ref = unet_torch.unet_forward(sd, x[i:i + 1].cpu()).numpy()
        assert _rel(y[i:i + 1].cpu().numpy(), ref) <= TOL


def test_unet_full_size_batch256_fp32(net_invariant, dev, weights_np):
    """north_star's batch and the per-rank shard of BASELINE configs[3] (2048 clips over 8 GPUs = 256 per GPU),
    fp32, 513x256: finite, clip independence bit-exact for three clips, two clips against the torch oracle."""
    from oracle import unet_torch
    n, f, t = 256, 513, 256
    g = torch.Generator(device=dev).manual_seed(0)                 # bench.py's rank-0 shard generator
    x = torch.rand((n, 1, f, t), generator=g, device=dev) * 4.0
    net = net_invariant
    with torch.no_grad():
        y = net(x)
        assert y.shape == x.shape and torch.isfinite(y).all()
        for i in (0, 129, 255):
            assert torch.equal(net(x[i:i + 1].clone())[0], y[i]), i
    sd = unet_torch.to_torch_state(weights_np)
    for i in (1, 254):
        ref = unet_torch.unet_forward(sd, x[i:i + 1].cpu()).numpy()
        assert _rel(y[i:i + 1].cpu().numpy(), ref) <= TOL
    net._workspace = None                                          # 37 GB: give it back before the next test
    torch.cuda.empty_cache()


@pytest.mark.parametrize("pinned,copy", [(False, True), (True, True), (True, False)])
def test_host_batches_pipeline_equals_plain_calls(dev, weights_np, pinned, copy):
    """UNet.forward_host_batches (copies on their own streams, double buffered) yields, batch for batch and bit for bit, what
    ``model(batch.cuda()).cpu()`` gives -- five batches with a short last one, pageable and pinned inputs, copied outputs and views."""
    from audiodenoiser_amd import UNet
    from audiodenoiser_amd.weights import make_input
    net = UNet()
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in weights_np.items()})
    net.eval().to(dev)
    sizes = (3, 3, 3, 3, 2)
    batches = [torch.from_numpy(make_input(90 + i, n, 40, 48)) for i, n in enumerate(sizes)]
    if pinned:
        batches = [b.pin_memory() for b in batches]
    with torch.no_grad():
        want = [net(b.to(dev)).cpu() for b in batches]
        got = []
        for out in net.forward_host_batches(iter(batches), copy=copy):
            assert not out.is_cuda and out.shape == want[len(got)].shape
            got.append(out.clone())                     # copy=False: the view is only good until the next advance
        assert list(net.forward_host_batches([])) == []
        one = list(net.forward_host_batches([batches[0]]))
        with pytest.raises(ValueError, match="larger than the first"):
            list(net.forward_host_batches([batches[4], batches[0]]))
        with pytest.raises(ValueError, match="every batch must be"):
            list(net.forward_host_batches([batches[0], torch.zeros(3, 1, 40, 64)]))
    assert len(got) == len(want) and len(one) == 1 and torch.equal(one[0], want[0])
    for g, w in zip(got, want):
        assert torch.equal(g, w)


def test_reference_test_py_call_shape_cpu_model_cpu_tensors(dev, weights_np, golden_dir, tmp_path, monkeypatch):
    """Literally the reference's inference caller (test.py:63-66,100,112-114,118-122) with PYTHONPATH=compat: the
    model is loaded with map_location='cpu' and never moved, the batch is a CPU tensor, the loss inputs are CPU
    tensors.  The mirror stages them on the current ROCm device, runs the HIP path and answers on the CPU."""
    import importlib
    import sys
    from oracle import loss_torch
    from audiodenoiser_amd.weights import make_input
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.syspath_prepend(os.path.join(root, "compat"))
    for name in ("model", "loss"):
        sys.modules.pop(name, None)
    UNet = importlib.import_module("model").UNet                               # test.py:8  from model import UNet
    CombinedPerceptualLoss = importlib.import_module("loss").CombinedPerceptualLoss   # test.py:9
    model_path = str(tmp_path / "unet_denoiser_white.pth")
    torch.save({k: torch.from_numpy(np.array(v)) for k, v in weights_np.items()}, model_path)
    noisy_spectrograms = np.concatenate([make_input(7, 1, 257, 188)[:, 0]] +
                                        [make_input(40 + i, 1, 257, 188)[:, 0] for i in range(4)])   # (5, 257, 188)
    clean_spectrograms = np.concatenate([make_input(50 + i, 1, 257, 188)[:, 0] for i in range(5)])

    model = UNet(in_channels=1, num_classes=1)                                                         # test.py:63
    model.load_state_dict(torch.load(model_path, map_location='cpu', weights_only=True))               # test.py:65
    model.eval()                                                                                       # test.py:66
    noisy_torch = torch.tensor(noisy_spectrograms, dtype=torch.float32).unsqueeze(1)                   # test.py:100
    with torch.no_grad():                                                                              # test.py:112
        denoised_torch = model(noisy_torch)                                                            # test.py:113
        denoised_spectrograms = denoised_torch.squeeze(1).cpu().numpy()                                # test.py:114
    criterion = CombinedPerceptualLoss()                                                               # test.py:117
    with torch.no_grad():
        denoised_torch = torch.tensor(denoised_spectrograms, dtype=torch.float32).unsqueeze(1)
        clean_torch = torch.tensor(clean_spectrograms, dtype=torch.float32).unsqueeze(1)
        total_loss, stft_loss, mel_loss, l1_loss = criterion(denoised_torch, clean_torch)              # test.py:122

    assert not noisy_torch.is_cuda and not any(p.is_cuda for p in model.parameters())                  # nothing was moved
    g = np.load(os.path.join(golden_dir, "unet_257x188.npz"))
    assert denoised_spectrograms.shape == (5, 257, 188)
    assert _rel(denoised_spectrograms[0], g["y"][0, 0]) <= TOL                     # the reference's own forward
    with torch.no_grad():                                                           # same clips as device tensors
        on_dev = model(noisy_torch.to(dev))
    assert on_dev.is_cuda and np.array_equal(on_dev.cpu().numpy()[:, 0], denoised_spectrograms)
    ref = loss_torch.per_clip(denoised_torch, clean_torch).numpy().mean(axis=0)
    got = np.array([total_loss.item(), stft_loss.item(), mel_loss.item(), l1_loss.item()])
    assert not total_loss.is_cuda and np.allclose(got, ref, rtol=2e-4)
    for name in ("model", "loss"):
        sys.modules.pop(name, None)


def test_c_abi_rejects_misaligned_buffers_and_keeps_the_current_device(net, dev):
    from audiodenoiser_amd import _lib
    L = _lib.load()
    x = torch.zeros(1, 1, 16, 16, device=dev)
    with torch.no_grad():
        net(x)
    need = ctypes.c_size_t()
    _lib.check(L.adn_unet_workspace_bytes(net._handle, 1, 16, 16, ctypes.byref(need)), "ws")
    ws = torch.empty(need.value + 64, dtype=torch.uint8, device=dev)
    y = torch.empty_like(x)
    before = torch.cuda.current_device()
    assert L.adn_unet_forward(net._handle, x.data_ptr(), y.data_ptr(), 1, 16, 16, ws.data_ptr() + 4, need.value, None) == 1
    assert b"aligned" in L.adn_last_error()
    assert L.adn_unet_forward(net._handle, x.data_ptr() + 2, y.data_ptr(), 1, 16, 16, ws.data_ptr(), need.value, None) == 1
    assert L.adn_unet_forward(net._handle, x.data_ptr(), y.data_ptr(), 1, 16, 16, ws.data_ptr() + 16, need.value, None) == 0
    torch.cuda.synchronize()
    assert torch.cuda.current_device() == before


@pytest.mark.parametrize("algo", ["direct", "winograd", "winograd_f2", "winograd_f4"])
def test_unet_both_conv_algorithms(dev, weights_np, golden_dir, algo, monkeypatch):
    """The 3x3 layers have three kernels: Winograd F(4x4,3x3) where its 32x32 tiles fit and F(2x2,3x3) elsewhere
    (default), and the direct implicit GEMM (ADN_CONV_ALGO=direct, read when the handle is created).
    ADN_WINO_TILE=2 / 4 pin one Winograd form for every plain 3x3 layer (4: also where the tiles overhang most of the
    image).  All must meet the same tolerance, on a shape with pads in both dimensions and on the reference's own
    test shape."""
    from audiodenoiser_amd.model import UNet
    from audiodenoiser_amd.weights import make_input
    monkeypatch.delenv("ADN_CONV_ALGO", raising=False)
    monkeypatch.delenv("ADN_WINO_TILE", raising=False)
    if algo == "direct":
        monkeypatch.setenv("ADN_CONV_ALGO", "direct")
    elif algo != "winograd":
        monkeypatch.setenv("ADN_WINO_TILE", algo[-1])
    m = UNet(1, 1)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in weights_np.items()}, strict=True)
    m = m.to(dev).eval()
    for (n, f, t) in ((2, 33, 47), (1, 257, 188)):
        g = np.load(os.path.join(golden_dir, f"unet_{f}x{t}.npz"))
        with torch.no_grad():
            y = m(torch.from_numpy(make_input(7, n, f, t)).to(dev)).cpu().numpy()
        assert _rel(y, g["y"]) <= TOL, (algo, f, t)


@pytest.mark.parametrize("n,f,t", [(3, 20, 36), (1, 48, 100), (2, 31, 16), (1, 16, 130), (2, 97, 70)])
def test_unet_f4_every_block_forced(dev, weights_np, n, f, t, monkeypatch):
    """F(4x4,3x3) forced onto every plain / pooled 3x3 layer (ADN_WINO_TILE=4), odd shapes: tile overhang in both
    dimensions, images smaller than one 32x32 tile, virtual pad + concat sources; every block output against the oracle."""
    import oracle
    from audiodenoiser_amd.model import UNet
    from audiodenoiser_amd.weights import make_input
    monkeypatch.delenv("ADN_CONV_ALGO", raising=False)
    monkeypatch.setenv("ADN_WINO_TILE", "4")
    m = UNet(1, 1)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in weights_np.items()}, strict=True)
    m = m.to(dev).eval()
    x = make_input(23, n, f, t)
    ref, rtaps = oracle.unet_forward(weights_np, x, acc64=True, want_taps=True)
    with torch.no_grad():
        y, taps = m(torch.from_numpy(x).to(dev), return_taps=True)
        y_plain = m(torch.from_numpy(x).to(dev))
    for name in oracle.TAP_NAMES:
        assert _rel(taps[name].cpu().numpy(), rtaps[name]) <= TOL, name
    assert _rel(y.cpu().numpy(), ref) <= TOL
    assert _rel(y_plain.cpu().numpy(), ref) <= TOL


def test_unet_weights_follow_state_dict_updates(dev, weights_np):
    """load_state_dict / in-place edits re-pack the weights (BatchNorm fold is redone)."""
    import oracle
    from audiodenoiser_amd.model import UNet
    from audiodenoiser_amd.weights import make_input, make_state_dict
    m = UNet().to(dev).eval()
    x = make_input(3, 1, 16, 16)
    sd2 = make_state_dict(99)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in weights_np.items()})
    with torch.no_grad():
        y1 = m(torch.from_numpy(x).to(dev)).cpu().numpy()
        m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd2.items()})
        y2 = m(torch.from_numpy(x).to(dev)).cpu().numpy()
        assert _rel(y1, oracle.unet_forward(weights_np, x, acc64=True)) <= TOL
        assert _rel(y2, oracle.unet_forward(sd2, x, acc64=True)) <= TOL
        m.out.bias.add_(1.0)
        y3 = m(torch.from_numpy(x).to(dev)).cpu().numpy()
    assert np.allclose(y3, y2 + 1.0, atol=1e-5)


def test_c_abi_error_paths(net, dev):
    from audiodenoiser_amd import _lib
    L = _lib.load()
    x = torch.zeros(1, 1, 16, 16, device=dev)
    with torch.no_grad():
        net(x)                                          # make sure a handle exists
    h = net._handle
    y = torch.empty_like(x)
    ws = torch.empty(16, dtype=torch.uint8, device=dev)
    rc = L.adn_unet_forward(h, x.data_ptr(), y.data_ptr(), 1, 16, 16, ws.data_ptr(), ws.numel(), None)
    assert rc == 3 and b"workspace" in L.adn_last_error()
    rc = L.adn_unet_forward(h, x.data_ptr(), y.data_ptr(), 1, 8, 16, ws.data_ptr(), ws.numel(), None)
    assert rc == 1
    with torch.no_grad(), pytest.raises(ValueError):
        net(torch.zeros(1, 1, 15, 64, device=dev))
    with torch.no_grad(), pytest.raises(ValueError):
        net(torch.zeros(1, 2, 16, 16, device=dev))


def test_config0_wav_to_spectrogram_to_forward(net, dev, weights_np):
    """BASELINE configs[0] end to end on the device: one 3 s @ 44.1 kHz clip -> STFT 1024/256 centred (513x517)
    -> loader rule (fp16 round trip, crop to 513x256) -> forward, batch 1.  (The reference's own clips are not
    shipped; a synthetic clip of the same shape stands in)  Each stage is checked against the
    oracle on the SAME input the next stage consumes, so fp16 rounding ties cannot leak between stages."""
    import oracle
    from oracle import unet_torch
    from audiodenoiser_amd.data_loader import quantize_pad_on_device
    from audiodenoiser_amd.stft import stft_magnitude
    clip = np.random.default_rng(0).uniform(-1, 1, 132300).astype(np.float32)
    mag = stft_magnitude(torch.from_numpy(clip).to(dev), 1024, 256, True)
    assert tuple(mag.shape) == (513, 517)
    assert _rel(mag.cpu().numpy(), oracle.stft_mag(clip, 1024, 256, True)) <= TOL
    x = quantize_pad_on_device(mag[None], (513, 256))
    assert np.array_equal(x.cpu().numpy()[0, 0], oracle.quantize_pad(mag.cpu().numpy(), (513, 256)))
    with torch.no_grad():
        y = net(x)
    ref = unet_torch.unet_forward(unet_torch.to_torch_state(weights_np), x.cpu()).numpy()
    assert y.shape == (1, 1, 513, 256) and _rel(y.cpu().numpy(), ref) <= TOL


def test_config0_real_audio_stage_by_stage(net, dev, weights_np, golden_dir):
    """BASELINE configs[0] on REAL audio: the bundled 3 s clip (tests/golden/real_audio_*.npz, frozen from
    /root/reference/data/test/noise/17480-2-0-24.wav) -> STFT 1024/256 centred -> loader rule -> forward.  Each stage
    against the oracle on the input the next stage consumes, and the loader + forward stages against
    config0_real_audio.npz, which the reference's data_loader.py + model.py produced (tools/make_golden.py)."""
    import oracle
    from audiodenoiser_amd.data_loader import quantize_pad_on_device
    from audiodenoiser_amd.stft import stft_magnitude
    fx = load_real_audio_fixture(golden_dir)
    g = load_real_audio_fixture(golden_dir, "config0_real_audio.npz")
    clip = fx["lr_sum_int16"].astype(np.float32) / np.float32(65536.0)
    mag = stft_magnitude(torch.from_numpy(clip).to(dev), 1024, 256, True)
    assert tuple(mag.shape) == (513, 517)
    assert _rel(mag.cpu().numpy(), oracle.stft_mag(clip, 1024, 256, True)) <= TOL
    x = quantize_pad_on_device(mag[None], (513, 256))
    xh = x.cpu().numpy()[0, 0]
    assert np.array_equal(xh, oracle.quantize_pad(mag.cpu().numpy(), (513, 256)))        # bit exact on the same input
    gx = g["x_f16"].astype(np.float32)
    # device STFT vs oracle STFT differ by ~1e-7 relative, so a few values land on the other side of an fp16 rounding
    # boundary: at most one fp16 ulp (2^-10 relative), and rarely
    diff = xh != gx
    assert diff.mean() < 2e-3 and np.all(np.abs(xh - gx)[diff] <= 2.0 ** -10 * np.abs(gx)[diff] + 6e-8)
    with torch.no_grad():
        y = net(torch.from_numpy(gx[None, None]).to(dev))                                 # the golden's own input
        y_chain = net(x)                                                                  # the device chain's input
    assert _rel(y.cpu().numpy()[0, 0], g["y"]) <= TOL
    assert _rel(y_chain.cpu().numpy()[0, 0], g["y"]) <= 2e-3          # fp16-ulp input flips propagate, bounded


def test_stft_silence_dc_and_gaps(dev, golden_dir):
    """Signals the synthetic noise tests never produce: digital silence (exact zeros out), a DC offset, and the real
    clip with a silent lead-in (frames inside the gap are exactly zero)."""
    import oracle
    from audiodenoiser_amd.stft import stft_magnitude
    fx = load_real_audio_fixture(golden_dir)
    clip = fx["lr_sum_int16"].astype(np.float32) / np.float32(65536.0)
    gapped = clip.copy()
    gapped[:30000] = 0.0
    batch = np.stack([np.zeros_like(clip), gapped, clip + np.float32(0.25), clip])
    m = stft_magnitude(torch.from_numpy(batch).to(dev), 1024, 256, True).cpu().numpy()
    assert not m[0].any() and not m[1][:, :100].any()
    for i in (1, 2, 3):
        assert _rel(m[i], oracle.stft_mag(batch[i], 1024, 256, True)) <= TOL
    dc = stft_magnitude(torch.full((8192,), 0.5, device=dev), 1024, 256, True).cpu().numpy()
    assert np.allclose(dc[0, 4:-4], 256.0, rtol=1e-5) and np.allclose(dc[1, 4:-4], 128.0, rtol=1e-5)
    assert dc[2:, 4:-4].max() < 1e-3
    for n_fft, hop in ((512, 128), (256, 64)):                       # the reference's own setting and a smaller one
        r = stft_magnitude(torch.from_numpy(clip[:24000]).to(dev), n_fft, hop, True).cpu().numpy()
        assert _rel(r, oracle.stft_mag(clip[:24000], n_fft, hop, True)) <= TOL


# ---------------------------------------------------------------------------------------------- STFT
STFT_CASES = [(16000, 512, 128, False), (24000, 512, 128, True), (132300, 1024, 256, True),
              (132300, 1024, 256, False), (5000, 2048, 512, True), (4096, 4096, 1024, False),
              (700, 64, 16, True), (1000, 128, 100, True), (3000, 256, 64, False), (1024, 1024, 256, False)]


@pytest.mark.parametrize("L,n_fft,hop,center", STFT_CASES)
def test_stft_matches_oracle(dev, L, n_fft, hop, center):
    import oracle
    from audiodenoiser_amd.stft import stft_magnitude
    from audiodenoiser_amd.weights import make_audio
    a = make_audio(3, 3, L)
    ref = oracle.stft_mag(a, n_fft, hop, center)
    got = stft_magnitude(torch.from_numpy(a).to(dev), n_fft, hop, center).cpu().numpy()
    assert got.shape == ref.shape
    assert _rel(got, ref) <= TOL


def test_stft_reference_helpers_numpy_in_numpy_out(dev):
    import oracle
    from audiodenoiser_amd.stft import audio_to_magnitude_spectrogram, audio_to_spectrogram
    from audiodenoiser_amd.weights import make_audio
    chunk = make_audio(5, 1, 16000)[0]       # 2 s @ 8 kHz train chunk (create_train_dataset.py:22-23)
    m = audio_to_magnitude_spectrogram(chunk)
    assert isinstance(m, np.ndarray) and m.dtype == np.float32 and m.shape == (257, 122)
    assert _rel(m, oracle.stft_mag(chunk, 512, 128, False)) <= TOL
    clip = make_audio(6, 1, 24000)[0]        # 3 s test clip (create_test_dataset.py:39)
    m = audio_to_spectrogram(clip)
    assert m.shape == (257, 188)
    assert _rel(m, oracle.stft_mag(clip, 512, 128, True)) <= TOL


def test_stft_known_answers(dev):
    from audiodenoiser_amd.stft import stft_magnitude
    n, hop = 1024, 256
    L = n * 4
    i = np.arange(L)
    k0, amp = 100, 0.7
    x = (amp * np.cos(2 * np.pi * k0 * i / n)).astype(np.float32)
    m = stft_magnitude(torch.from_numpy(x).to(dev), n, hop, False).cpu().numpy()
    assert np.allclose(m[k0], amp * n / 4, rtol=1e-4) and np.allclose(m[k0 - 1], amp * n / 8, rtol=1e-4)
    rest = np.delete(m, [k0 - 1, k0, k0 + 1], axis=0)
    assert rest.max() < 1e-3 * amp * n / 4
    imp = np.zeros(L, np.float32)
    imp[300] = 1.0
    m = stft_magnitude(torch.from_numpy(imp).to(dev), n, hop, False).cpu().numpy()
    assert np.allclose(m[:, 0], 0.5 - 0.5 * np.cos(2 * np.pi * 300 / n), atol=1e-6)
    assert np.allclose(m[:, 1], 0.5 - 0.5 * np.cos(2 * np.pi * 44 / n), atol=1e-6)   # sample 300 is index 44 of frame 1


def test_stft_full_size_properties(dev):
    """BASELINE config 3 shape (132300-sample clips, 1024/256, centred) at 2000 clips: exact homogeneity,
    clip independence, and three clips against the oracle."""
    import oracle
    from audiodenoiser_amd.stft import stft_magnitude
    n_clips, L = 2000, 132300
    g = torch.Generator(device=dev).manual_seed(0)
    a = torch.rand((n_clips, L), generator=g, device=dev) * 2 - 1
    m = stft_magnitude(a, 1024, 256, True)
    assert m.shape == (n_clips, 513, 517) and torch.isfinite(m).all()
    assert torch.equal(stft_magnitude(a * 2.0, 1024, 256, True), m * 2.0)          # scaling by 2 is exact
    for i in (0, 777, n_clips - 1):
        assert torch.equal(stft_magnitude(a[i], 1024, 256, True), m[i])
        assert _rel(m[i].cpu().numpy(), oracle.stft_mag(a[i].cpu().numpy(), 1024, 256, True)) <= TOL


def test_stft_config2_full_size_10000_clips(dev):
    """BASELINE configs[2] at its stated size: 10 000 clips x 132 300 samples (5.3 GB in, 10.6 GB out), n_fft 1024,
    hop 256, centred: exact homogeneity, clip independence, three clips against the oracle."""
    import oracle
    from audiodenoiser_amd.stft import stft_magnitude
    n_clips, L = 10000, 132300
    g = torch.Generator(device=dev).manual_seed(0)
    a = torch.rand((n_clips, L), generator=g, device=dev) * 2 - 1
    m = stft_magnitude(a, 1024, 256, True)
    assert m.shape == (n_clips, 513, 517)
    assert bool(torch.isfinite(m).all())
    m2 = stft_magnitude(a * 2.0, 1024, 256, True)
    m2 *= 0.5                                                        # scaling by a power of two is exact
    assert torch.equal(m2, m)
    del m2
    for i in (0, 4999, n_clips - 1):
        assert torch.equal(stft_magnitude(a[i], 1024, 256, True), m[i])
        assert _rel(m[i].cpu().numpy(), oracle.stft_mag(a[i].cpu().numpy(), 1024, 256, True)) <= TOL
    del m, a
    torch.cuda.empty_cache()


@pytest.mark.parametrize("L,n_fft,hop,center,target", [
    (132300, 1024, 256, True, (513, 256)),      # BASELINE configs[0]: crop 517 -> 256 frames
    (24000, 512, 128, True, (256, 64)),         # the reference's loader default on its test clips: crop both ways
    (16000, 512, 128, False, (257, 160)),       # train chunk (257 x 122) padded on the right
    (5000, 256, 64, True, (200, 100)),          # pad rows (129 < 200) and frames (79 < 100)
    (6000, 2048, 512, True, (513, 8)),          # workgroup-synchronous kernel (n_fft > 1024), crop rows and frames
    (3000, 64, 16, True, (33, 188)),
    # the persistent whole-line kernel (n_fft 256 / 512 / 1024): windows that are not multiples of 4 or 32 frames (scalar
    # store tail), odd clip lengths (clips alternate 8-byte alignment), a hop that is not n_fft / 4, fewer rows than bins
    (132301, 1024, 256, True, (513, 250)),
    (40001, 1024, 200, True, (300, 131)),
    (24001, 512, 128, True, (257, 67)),
    (9000, 256, 64, False, (129, 97)),
    (20000, 512, 100, False, (100, 33)),
])
def test_stft_mag_fit_equals_stft_then_loader_rule(dev, L, n_fft, hop, center, target):
    """adn_stft_mag_fit = adn_quantize_pad(adn_stft_mag(...)) in one kernel, bit for bit (it only skips the frames and
    rows outside the window), and within fp16 rounding of the CPU oracle chain."""
    import oracle
    from audiodenoiser_amd.data_loader import quantize_pad_on_device
    from audiodenoiser_amd.stft import stft_magnitude, stft_magnitude_fit
    from audiodenoiser_amd.weights import make_audio
    a = make_audio(9, 3, L) * np.float32(0.5)
    ad = torch.from_numpy(a).to(dev)
    fused = stft_magnitude_fit(ad, target, n_fft, hop, center)
    two_step = quantize_pad_on_device(stft_magnitude(ad, n_fft, hop, center), target)
    assert fused.shape == (3, 1) + tuple(target) and torch.equal(fused, two_step)
    ref = oracle.quantize_pad(oracle.stft_mag(a[1], n_fft, hop, center), target)
    got = fused[1, 0].cpu().numpy()
    assert np.max(np.abs(got - ref)) <= 2.0 ** -10 * np.max(np.abs(ref))       # one fp16 ulp where roundings straddle
    assert (got != ref).mean() < 5e-3


def test_stft_mag_fit_persistent_runs_many_clips(dev):
    """The persistent kernel of adn_stft_mag_fit at a size where every workgroup walks a run of several (clip, group) items
    that crosses clip boundaries: 700 clips x 8 groups over 768 resident workgroups.  Bit-identical to the two-step form, and
    every clip independent of its neighbours (a clip computed alone gives the same bits)."""
    from audiodenoiser_amd.data_loader import quantize_pad_on_device
    from audiodenoiser_amd.stft import stft_magnitude, stft_magnitude_fit
    g = torch.Generator(device=dev).manual_seed(4)
    a = torch.rand((700, 132300), generator=g, device=dev) * 2 - 1
    fused = stft_magnitude_fit(a, (513, 256), 1024, 256, True)
    for lo in (0, 350):
        two_step = quantize_pad_on_device(stft_magnitude(a[lo:lo + 350], 1024, 256, True), (513, 256))
        assert torch.equal(fused[lo:lo + 350], two_step)
        del two_step
    for i in (0, 1, 313, 699):
        assert torch.equal(stft_magnitude_fit(a[i:i + 1].clone(), (513, 256), 1024, 256, True)[0], fused[i])
    # the reference's own setting (n_fft 512 / hop 128, 3 s @ 8 kHz -> 257 x 188 cropped to the loader default 256 x 64)
    b = torch.rand((3000, 24000), generator=g, device=dev) * 2 - 1
    f2 = stft_magnitude_fit(b, (256, 64), 512, 128, True)
    assert torch.equal(f2, quantize_pad_on_device(stft_magnitude(b, 512, 128, True), (256, 64)))


def test_bench_two_rank_rehearsal_on_one_gpu(dev):
    """The driver's N > 1 command line (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`) end to end with
    two ranks, both on this GPU, gloo moving the CUDA tensors (ADN_BENCH_REHEARSAL=1: a rehearsal, not a measurement): the ranks
    shard the clips, the per-clip values are gathered, rank 0 prints ONE JSON line that says how many ranks the collective held."""
    import json
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ADN_BENCH_REHEARSAL="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch-per-gpu", "4"], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["unit"] == "frames/s"
    assert d["ranks"]["ranks_seen"] == 2 and d["ranks"]["rank_ids"] == [0, 1] and d["ranks"]["allgather_ms"] > 0
    assert d["config"]["batch_per_gpu"] == 4 and d["config"]["global_batch"] == 8
    assert abs(d["value"] - 8 * 256 * 2 / (d["ms_per_step"] * 2e-3)) <= 0.01 * d["value"]
    assert "REHEARSAL" in d["data"] and "roofline" in d and "stft" not in d        # sub-benchmarks are N = 1 only


def test_stft_rejects_bad_arguments(dev):
    from audiodenoiser_amd._lib import AdnError
    from audiodenoiser_amd.stft import stft_magnitude
    a = torch.zeros(1000, device=dev)
    with pytest.raises(ValueError):
        stft_magnitude(a, 2048, 512, False)
    with pytest.raises(AdnError):
        stft_magnitude(a, 500, 128, True)
    with pytest.raises(RuntimeError):
        stft_magnitude(torch.zeros(1000), 512, 128, True)


# ---------------------------------------------------------------------------------------------- loader / loss
def test_quantize_pad_bit_exact(dev, golden_dir):
    from audiodenoiser_amd.data_loader import quantize_pad_on_device
    from audiodenoiser_amd.weights import hash_uniform
    g = np.load(os.path.join(golden_dir, "loader_cases.npz"))
    for ci in range(4):
        shape = tuple(g[f"case{ci}_in_shape"])
        target = tuple(g[f"case{ci}_target"])
        u = hash_uniform(5, f"loader{ci}", 2 * shape[0] * shape[1]).reshape(2, *shape)
        noisy = (u[0] * np.float32(8.0)).astype(np.float32)
        clean = (u[1] * np.float32(8.0)).astype(np.float32)
        noisy[0, 0], noisy[0, 1], noisy[0, 2], noisy[1, 0] = 70000.0, 1e-8, 3e-6, 65504.0
        out = quantize_pad_on_device(torch.from_numpy(np.stack([noisy, clean])).to(dev), target).cpu().numpy()
        assert out.shape == (2, 1) + target
        assert np.array_equal(out[0], g[f"case{ci}_noisy"]) and np.array_equal(out[1], g[f"case{ci}_clean"])


def test_quantize_pad_random_bit_patterns(dev):
    from audiodenoiser_amd.data_loader import quantize_pad_on_device
    bits = np.random.default_rng(0).integers(0, 2 ** 32, size=(1, 512, 512), dtype=np.uint64).astype(np.uint32)
    x = bits.view(np.float32)
    x = np.where(np.isnan(x), np.float32(1.0), x)
    with np.errstate(over="ignore"):
        ref = x.astype(np.float16).astype(np.float32)
    out = quantize_pad_on_device(torch.from_numpy(x).to(dev), (512, 512)).cpu().numpy()
    assert np.array_equal(out[:, 0], ref)


def test_per_clip_l1(dev):
    from audiodenoiser_amd.loss import per_clip_l1
    g = torch.Generator().manual_seed(0)
    a = torch.rand((5, 1, 33, 47), generator=g)
    b = torch.rand((5, 1, 33, 47), generator=g)
    got = per_clip_l1(a.to(dev), b.to(dev)).cpu().numpy()
    ref = (a.double() - b.double()).abs().reshape(5, -1).mean(dim=1).numpy()
    assert np.allclose(got, ref, rtol=1e-5)


@pytest.mark.parametrize("b,f,t", [(3, 40, 96), (2, 257, 188), (4, 513, 256), (1, 33, 64), (2, 40, 32), (3, 24, 48), (2, 16, 63)])
def test_perceptual_loss_per_clip(dev, b, f, t):
    """Per-clip CombinedPerceptualLoss (loss.py:6-95) against the torch oracle; tolerance 1e-4 relative per term.
    (Mel term: parity unpinned upstream, torchaudio absent — oracle restates its published defaults.)"""
    from oracle import loss_torch
    from audiodenoiser_amd.loss import CombinedPerceptualLoss, perceptual_loss_per_clip
    g = torch.Generator().manual_seed(5)
    pred = torch.rand((b, 1, f, t), generator=g) * 3
    tgt = torch.rand((b, 1, f, t), generator=g) * 3
    ref = loss_torch.per_clip(pred, tgt).numpy()
    got = perceptual_loss_per_clip(pred.to(dev), tgt.to(dev)).cpu().numpy()
    assert got.shape == (b, 4)
    assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max(), (got, ref)
    assert np.allclose(got, ref, rtol=2e-4, atol=1e-6)
    total, stft, mel, l1 = CombinedPerceptualLoss()(pred.to(dev), tgt.to(dev))
    assert abs(float(l1) - float(torch.nn.functional.l1_loss(pred, tgt))) < 1e-5
    assert abs(float(total) - float(ref[:, 0].mean())) < 1e-4 * float(ref[:, 0].mean())


@pytest.mark.parametrize("t", [2688, 2704, 4094, 6784, 6785, 8192, 20001, 65535])
def test_perceptual_loss_long_clips(dev, t):
    """Frame counts beyond 64 KiB of LDS per clip (T >= 2689), up to what one CU's LDS holds (6784), and beyond it (the series
    then live in the workspace and the mel frames are walked in blocks): the reference's loss has no length limit."""
    from oracle import loss_torch
    from audiodenoiser_amd.loss import perceptual_loss_per_clip
    g = torch.Generator().manual_seed(t)
    pred = torch.rand((2, 1, 24, t), generator=g) * 3
    tgt = torch.rand((2, 1, 24, t), generator=g) * 3
    ref = loss_torch.per_clip(pred, tgt).numpy()
    got = perceptual_loss_per_clip(pred.to(dev), tgt.to(dev)).cpu().numpy()
    assert np.allclose(got, ref, rtol=2e-4, atol=1e-6)


def test_perceptual_loss_zero_and_errors(dev):
    from audiodenoiser_amd.loss import perceptual_loss_per_clip
    x = torch.rand((2, 1, 64, 128), device=dev)
    assert float(perceptual_loss_per_clip(x, x.clone()).abs().max()) == 0.0
    with pytest.raises(ValueError):
        perceptual_loss_per_clip(x, x[:, :, :32])
    from audiodenoiser_amd._lib import AdnError
    with pytest.raises(AdnError):
        perceptual_loss_per_clip(x[..., :31].contiguous(), x[..., :31].contiguous())     # T < 32: reflect pad 31 needs T > 31


def test_convt_split_bf16_matches_exact_fp32_form(dev, weights_np, golden_dir, monkeypatch):
    """The fp32 transposed convolutions run on the bf16 matrix cores through a three-term split of both operands (six products,
    fp32 accumulation).  That is fp32-level arithmetic: against the exact-fp32 MFMA form (ADN_CONVT_SPLIT=0, read when a handle is
    created) the whole network differs by what any re-ordering of fp32 sums gives after 23 layers -- a few 1e-6 of max|y| (the
    F(4x4,3x3) and F(2x2,3x3) forms of the SAME library differ by as much) -- on even and odd shapes, and both forms sit at the
    same distance from the reference golden."""
    from audiodenoiser_amd.model import UNet
    from audiodenoiser_amd.weights import make_input

    def make():
        m = UNet(1, 1)
        m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in weights_np.items()}, strict=True)
        return m.to(dev).eval()

    split = make()
    monkeypatch.setenv("ADN_CONVT_SPLIT", "0")
    exact = make()
    with torch.no_grad():
        exact(torch.zeros((1, 1, 16, 16), device=dev))             # the handle (and its switch) is created at the first forward
    monkeypatch.delenv("ADN_CONVT_SPLIT")
    for (n, f, t) in ((2, 33, 47), (1, 257, 188), (1, 513, 256)):
        x = torch.from_numpy(make_input(7, n, f, t)).to(dev)
        g = np.load(os.path.join(golden_dir, f"unet_{f}x{t}.npz"))
        with torch.no_grad():
            ys, taps_s = split(x, return_taps=True)
            ye, taps_e = exact(x, return_taps=True)
        scale = float(ye.abs().max())
        assert float((ys - ye).abs().max()) <= 2e-5 * scale, (f, t)
        assert not torch.equal(ys, ye) or f < 64                # (different arithmetic: identical bits would mean the switch did nothing)
        for name in ("up1", "up2", "up3", "up4"):
            d = float((taps_s[name] - taps_e[name]).abs().max()) / float(taps_e[name].abs().max())
            assert d <= 2e-5, (name, d)
        es, ee = _rel(ys.cpu().numpy(), g["y"]), _rel(ye.cpu().numpy(), g["y"])
        assert es <= TOL and ee <= TOL and es <= 2.0 * ee + 1e-6


# ---------------------------------------------------------------------------------------------- fp16 path
@pytest.mark.parametrize("conv", ["default", "first0", "32", "convt_dma"])
def test_fp16_path_within_1e2_of_fp32_reference(dev, weights_np, golden_dir, conv, monkeypatch):
    """BASELINE configs[4]: fp16 storage + fp16 MFMA (fp32 accumulate); outputs within 1e-2 (relative to max|ref|)
    of the fp32 reference goldens, on every golden shape, and block outputs within 1e-2 of their rms.  Both 3x3 kernel
    families: conv16_f16 (16x16x32 MFMA; the default, with Conv2d(1 -> 64) computed inside down1's second conv, and with
    ADN_F16_FIRST=0 as its own launch: down1's second conv then runs the resident-weight pooling form) and
    conv_dma<_Float16> (32x32x16, ADN_F16_CONV=32); and both transposed-convolution kernels: convt16_f16 (the default) and
    conv_dma<_Float16, ..., CONVT2X2> (ADN_F16_CONVT=dma)."""
    from audiodenoiser_amd.model import UNet
    from audiodenoiser_amd.weights import make_input
    for k in ("ADN_F16_CONV", "ADN_F16_FIRST", "ADN_F16_CONVT"):
        monkeypatch.delenv(k, raising=False)
    if conv == "32":
        monkeypatch.setenv("ADN_F16_CONV", conv)
    elif conv == "first0":
        monkeypatch.setenv("ADN_F16_FIRST", "0")
    elif conv == "convt_dma":
        monkeypatch.setenv("ADN_F16_CONVT", "dma")
    m = UNet(1, 1)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in weights_np.items()}, strict=True)
    m = m.to(dev).eval().set_compute_dtype("f16")
    with torch.no_grad():
        m(torch.zeros((1, 1, 16, 16), device=dev))               # the handle (and its switches) is created at the first forward
    for k in ("ADN_F16_CONV", "ADN_F16_FIRST", "ADN_F16_CONVT"):
        monkeypatch.delenv(k, raising=False)
    for (n, f, t) in GOLDEN_SHAPES + ((3, 20, 36), (2, 31, 16)):
        if (f, t) not in [(s[1], s[2]) for s in GOLDEN_SHAPES]:
            import oracle
            x = make_input(21, n, f, t)
            ref = oracle.unet_forward(weights_np, x, acc64=True)
            with torch.no_grad():
                y, y_taps = m(torch.from_numpy(x).to(dev)), m(torch.from_numpy(x).to(dev), return_taps=True)[0]
            assert _rel(y.cpu().numpy(), ref) <= 1e-2 and _rel(y_taps.cpu().numpy(), ref) <= 1e-2, (f, t)
            continue
        g = np.load(os.path.join(golden_dir, f"unet_{f}x{t}.npz"))
        with torch.no_grad():
            y, taps = m(torch.from_numpy(make_input(7, n, f, t)).to(dev), return_taps=True)
            y_plain = m(torch.from_numpy(make_input(7, n, f, t)).to(dev))         # production sequence: fused 1x1 tail
        assert y.dtype == torch.float32
        assert _rel(y.cpu().numpy(), g["y"]) <= 1e-2, (f, t)
        assert _rel(y_plain.cpu().numpy(), g["y"]) <= 1e-2, (f, t)
        for name, tp in taps.items():
            a = tp.cpu().numpy().astype(np.float64).ravel()
            s_, sa, sq, cnt = g[f"{name}_stats"]
            rms = np.sqrt(sq / cnt)
            assert np.abs(a[g[f"{name}_idx"]] - g[f"{name}_val"]).max() <= 5e-2 * rms, name
            assert abs(np.abs(a).sum() - sa) <= 1e-2 * sa, name


def test_fp16_path_batch256(dev, weights_np):
    """configs[4] at its stated size, batch 256 x 513x256: finite, clip independent, and two clips within 1e-2 of
    the oracle's fp32 forward (oracle/unet_torch.py: the ATen kernels the reference dispatches to)."""
    from oracle import unet_torch
    from audiodenoiser_amd.model import UNet
    m = UNet(1, 1)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in weights_np.items()}, strict=True)
    m = m.to(dev).eval().set_compute_dtype("f16")
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.rand((256, 1, 513, 256), generator=g, device=dev) * 4.0
    with torch.no_grad():
        y = m(x)
        assert torch.isfinite(y).all()
        # default handle: a clip alone runs its deep layers as K-split slices (another summation order): close, not bit-equal
        alone = m(x[100:101].clone())
        assert _rel(alone.cpu().numpy(), y[100:101].cpu().numpy()) <= 5e-3
        m.set_batch_invariant(True)                          # one kernel per layer whatever the batch: bit-equal
        yi = m(x)
        assert torch.equal(m(x[100:101].clone())[0], yi[100])
        m.set_batch_invariant(False)
    sd = unet_torch.to_torch_state(weights_np)
    for i in (0, 255):
        ref = unet_torch.unet_forward(sd, x[i:i + 1].cpu()).numpy()
        assert _rel(y[i:i + 1].cpu().numpy(), ref) <= 1e-2, i
    assert _rel(alone.cpu().numpy(), unet_torch.unet_forward(sd, x[100:101].cpu()).numpy()) <= 1e-2
    m._workspace = None
    torch.cuda.empty_cache()


@pytest.mark.gpu
def test_wav_to_spec_dataset_matches_oracle(dev, tmp_path):
    """On-the-fly wav dataset (train.py:106-109): item = fp32(fp16(|STFT|)) cropped/padded, vs the CPU oracle."""
    from audiodenoiser_amd.data_loader import WavToSpecDataset
    import oracle
    from audiodenoiser_amd.wav import read_wav, write_wav
    rng = np.random.default_rng(11)
    lengths = [8000, 8000, 6000, 24000]
    for i, n in enumerate(lengths):
        clean = (rng.uniform(-1, 1, n) * 0.3).astype(np.float32)
        noisy = (clean + rng.normal(0, 0.05, n)).astype(np.float32)
        write_wav(str(tmp_path / f"clean_{i}.wav"), clean, 8000, "FLOAT" if i % 2 else "PCM_16")
        write_wav(str(tmp_path / f"noisy_{i}.wav"), noisy, 8000, "FLOAT" if i % 2 else "PCM_16")
    ds = WavToSpecDataset(str(tmp_path), target_size=(256, 64), sample_rate=8000, device=dev)
    assert len(ds) == 4
    for i in range(4):
        noisy, clean = ds[i]
        assert noisy.shape == clean.shape == (1, 256, 64) and noisy.dtype == torch.float32 and not noisy.is_cuda
        for got, name in ((noisy, "noisy"), (clean, "clean")):
            audio, _ = read_wav(str(tmp_path / f"{name}_{i}.wav"))
            ref = oracle.quantize_pad(oracle.stft_mag(audio, 512, 128, True), (256, 64))
            # fp16 rounding of values that differ by 1e-6 relative may flip one fp16 ulp (2^-11 relative)
            assert np.max(np.abs(got.numpy()[0] - ref)) <= 2.0 ** -10 * np.max(np.abs(ref))
            assert _rel(got.numpy()[0], ref) <= 1e-3
    nb, cb = ds.load_batch_to_device([0, 1])
    assert nb.shape == (2, 1, 256, 64) and nb.is_cuda
    assert torch.equal(nb[0].cpu(), ds[0][0]) and torch.equal(cb[1].cpu(), ds[1][1])
    with pytest.raises(ValueError):
        WavToSpecDataset(str(tmp_path), sample_rate=16000, device=dev)[0]
    # the DataLoader feed of train.py:118-119: workers decode audio only (host), the main process transforms on the device
    from torch.utils.data import DataLoader
    # clip_samples = the 8320 samples the 64 frames reach: every item equals ds[i] although the loader crops / pads the AUDIO
    # and ds[i] the spectrogram -- files of 8000 and 6000 samples (shorter: zero padded) and of 24000 (longer: cropped)
    assert ds.min_clip_samples() == 8320
    with pytest.raises(ValueError, match="8320"):
        ds.loader(8000, batch_size=2)
    loader = ds.loader(8320, batch_size=2, num_workers=2, pin_memory=True)
    assert len(loader) == 2
    batches = list(loader)
    assert len(batches) == 2 and all(b[0].is_cuda and b[0].shape == (2, 1, 256, 64) for b in batches)
    assert torch.equal(batches[0][0], nb) and torch.equal(batches[0][1], cb)
    for i in range(4):
        assert torch.equal(batches[i // 2][0][i % 2].cpu(), ds[i][0]) and torch.equal(batches[i // 2][1][i % 2].cpu(), ds[i][1])
    # a shorter crop is allowed on request; the 24000-sample file then differs from ds[3] in the frames that reach past it
    cut = list(ds.loader(8000, allow_cut_frames=True, batch_size=4))[0][0]
    assert torch.equal(cut[0].cpu(), ds[0][0]) and not torch.equal(cut[3].cpu(), ds[3][0])
    assert torch.equal(cut[3, :, :, :60].cpu(), ds[3][0][:, :, :60])
    # train.py:111-119 as written: random_split, then a loader per split
    from torch.utils.data import random_split
    tr_split, va_split = random_split(ds, [3, 1], generator=torch.Generator().manual_seed(1))
    vb = list(ds.loader(8320, subset=va_split, batch_size=2))
    assert len(vb) == 1 and torch.equal(vb[0][0][0].cpu(), ds[va_split.indices[0]][0])
    # DataLoader(ds, num_workers > 0) itself: a worker forked from this GPU-initialised process cannot use HIP -> clear error;
    # a spawned worker opens its own context and works
    with pytest.raises(RuntimeError, match="DataLoader worker"):
        next(iter(DataLoader(ds, batch_size=2, num_workers=1)))
    sp = next(iter(DataLoader(ds, batch_size=2, num_workers=1, multiprocessing_context="spawn")))
    assert torch.equal(sp[0][0], ds[0][0]) and torch.equal(sp[1][1], ds[1][1])


@pytest.mark.gpu
@pytest.mark.parametrize("n_fft,hop,nfr", [(512, 128, 188), (256, 64, 33), (1024, 256, 40), (512, 256, 21), (64, 16, 9)])
def test_istft_and_complex_stft_match_oracle(dev, n_fft, hop, nfr):
    from oracle import griffin_lim_numpy as gl
    from audiodenoiser_amd.griffin_lim import istft, stft_complex
    rng = np.random.default_rng(n_fft + nfr)
    spec = (rng.normal(size=(2, n_fft // 2 + 1, nfr)) + 1j * rng.normal(size=(2, n_fft // 2 + 1, nfr))).astype(np.complex64)
    fm = torch.from_numpy(np.ascontiguousarray(spec.transpose(0, 2, 1))).to(dev)        # frame-major
    got = istft(fm, hop).cpu().numpy()
    for c in range(2):
        ref = gl.istft(spec[c], hop)
        assert got[c].shape == ref.shape
        assert np.max(np.abs(got[c] - ref)) <= TOL * np.max(np.abs(ref))
    audio = rng.uniform(-1, 1, (2, hop * (nfr - 1))).astype(np.float32)
    z = stft_complex(torch.from_numpy(audio).to(dev), n_fft, hop).cpu().numpy()
    for c in range(2):
        ref = gl.stft_complex(audio[c], n_fft, hop, True).T
        assert z[c].shape == ref.shape
        assert np.max(np.abs(z[c] - ref)) <= TOL * np.max(np.abs(ref))
    # round trip on the device: Hann at hop <= n_fft/2 reconstructs exactly
    back = istft(stft_complex(torch.from_numpy(audio).to(dev), n_fft, hop), hop).cpu().numpy()
    assert np.max(np.abs(back - audio)) <= 1e-5


@pytest.mark.gpu
def test_griffin_lim_matches_oracle_and_reference_shapes(dev):
    """test.py:29-48 with the reference's sizes (257 x 188, n_fft 512, hop 128, 50 iterations)."""
    from oracle import griffin_lim_numpy as gl
    from audiodenoiser_amd.griffin_lim import griffin_lim_reconstruction
    rng = np.random.default_rng(8)
    clip = rng.uniform(-1, 1, 24000).astype(np.float32)
    mag = np.abs(gl.stft_complex(clip, 512, 128, True)).astype(np.float32)            # (257, 188)
    rand = rng.random(mag.shape)
    for iters in (0, 2, 50):
        got = griffin_lim_reconstruction(mag, 512, 128, iterations=iters, rand=rand)
        assert isinstance(got, np.ndarray) and got.shape == (128 * 187,) and got.dtype == np.float32
        ref = gl.griffin_lim(mag, 512, 128, min(iters, 2), rand)     # the loop is a fixed point (see oracle test)
        assert np.max(np.abs(got - ref)) <= TOL * np.max(np.abs(ref))
    # batch + tensor in / tensor out, unseeded start phase like the reference
    mags = torch.from_numpy(np.stack([mag, 0.5 * mag])).to(dev)
    out = griffin_lim_reconstruction(mags, 512, 128, iterations=3)
    assert out.is_cuda and out.shape == (2, 128 * 187) and torch.isfinite(out).all()
    # statistical check for the random start: projecting a random-phase spectrogram onto the consistent ones keeps
    # about hop/n_fft = 1/4 of its energy (4x redundant frames)
    z = np.abs(gl.stft_complex(out[0].cpu().numpy(), 512, 128, True))
    assert 0.15 < float(np.sum(z ** 2) / np.sum(mag ** 2)) < 0.4
    from audiodenoiser_amd._lib import AdnError
    with pytest.raises(AdnError):
        griffin_lim_reconstruction(mag[:200], 512, 128)


@pytest.mark.gpu
def test_end_to_end_test_script_mirror(dev, tmp_path):
    """tools/run_test_set.py = the reference's test.py flow (npy test set -> forward -> losses -> Griffin-Lim -> wav +
    metrics files) on fabricated data; checks the artefacts test.py would leave behind."""
    import subprocess
    import sys
    from audiodenoiser_amd.wav import read_wav
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    data, models, out = (str(tmp_path / d) for d in ("data", "models", "out"))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "run_test_set.py"), "--synthetic", "--data", data,
                        "--models", models, "--out", out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "processed noise types: ['white', 'urban']" in r.stdout
    for nt in ("white", "urban"):
        txt = open(os.path.join(out, f"{nt}_metrics.txt")).read()
        assert txt.startswith(f"Perceptual metrics for noise type '{nt}':") and "Total Loss:" in txt and "L1 Loss:" in txt
        audio, rate = read_wav(os.path.join(out, f"{nt}_denoised_0.wav"))
        assert rate == 8000 and audio.shape == (128 * 187,) and np.isfinite(audio).all()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_forward_is_stream_capturable(net, weights_np, dev, dtype):
    """The launch sequence only enqueues on the caller's stream (no allocation, no synchronisation), so it can be
    captured into a HIP graph and replayed — what a serving loop does to drop the 23 launch overheads at batch 1.
    Both arithmetic types (fp16: the persistent conv16_f16 / convt16_f16 kernels)."""
    if dtype == "f16":
        from audiodenoiser_amd.model import UNet
        net = UNet(1, 1)
        net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in weights_np.items()}, strict=True)
        net = net.to(dev).eval().set_compute_dtype("f16")
    x = torch.rand((1, 1, 64, 48), device=dev) * 3
    with torch.no_grad():
        ref = net(x).clone()                       # warm-up: packs weights, sizes the workspace
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            net(x)
        torch.cuda.current_stream(dev).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            y = net(x)
        x.copy_(torch.rand((1, 1, 64, 48), device=dev) * 3)
        graph.replay()
        torch.cuda.synchronize()
        again = net(x)
    assert torch.equal(y, again) and not torch.equal(y, ref)


@pytest.mark.gpu
def test_wav_to_network_chain_is_graph_capturable_after_prepare(net, dev):
    """adn_prepare(device, n_fft) builds the constant tables ahead of time; after it adn_stft_mag_fit + adn_unet_forward (the
    wav -> network path of test.py:94-113) only enqueue, so the chain is recorded into one HIP graph and replayed."""
    from audiodenoiser_amd.stft import prepare, stft_magnitude_fit
    prepare(dev, 512)
    audio = torch.rand((2, 24000), device=dev) * 2 - 1
    with torch.no_grad():
        ref = net(stft_magnitude_fit(audio, (256, 64), 512, 128, True)).clone()      # warm-up (workspace, weights)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            net(stft_magnitude_fit(audio, (256, 64), 512, 128, True))
        torch.cuda.current_stream(dev).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            y = net(stft_magnitude_fit(audio, (256, 64), 512, 128, True))
        audio.copy_(torch.rand((2, 24000), device=dev) * 2 - 1)
        graph.replay()
        torch.cuda.synchronize()
        again = net(stft_magnitude_fit(audio, (256, 64), 512, 128, True))
    assert torch.equal(y, again) and not torch.equal(y, ref)


@pytest.mark.gpu
def test_cold_table_lookup_on_a_capturing_stream_is_refused_cleanly(dev):
    """In a fresh process (cold tables): adn_stft_mag and adn_perceptual_loss issued on a capturing stream return ADN_ERR_INVALID
    with a message naming adn_prepare -- no HIP error, nothing enqueued, the capture stays valid; after adn_prepare the same calls
    are recorded."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import ctypes, torch
from audiodenoiser_amd import _lib
L = _lib.load()
dev = torch.device("cuda", 0)
a = torch.rand((1, 4096), device=dev)
out = torch.empty((1, 129, 65), device=dev)
p = torch.rand((1, 1, 40, 64), device=dev); q = torch.rand((1, 1, 40, 64), device=dev)
need = ctypes.c_size_t()
assert L.adn_perceptual_loss_workspace_bytes(1, 40, 64, ctypes.byref(need)) == 0
ws = torch.empty(need.value, dtype=torch.uint8, device=dev); lo = torch.empty((1, 4), device=dev)
torch.cuda.synchronize()
s = torch.cuda.Stream(device=dev)
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    g.capture_begin()
    rc1 = L.adn_stft_mag(a.data_ptr(), 1, 4096, 256, 64, 1, out.data_ptr(), s.cuda_stream)
    m1 = L.adn_last_error()
    rc2 = L.adn_perceptual_loss(p.data_ptr(), q.data_ptr(), 1, 40, 64, ws.data_ptr(), need.value, lo.data_ptr(), s.cuda_stream)
    m2 = L.adn_last_error()
    g.capture_end()
assert rc1 == 1 and b"adn_prepare" in m1, (rc1, m1)
assert rc2 == 1 and b"adn_prepare" in m2, (rc2, m2)
assert L.adn_prepare(0, 256) == 0
g2 = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    g2.capture_begin()
    assert L.adn_stft_mag(a.data_ptr(), 1, 4096, 256, 64, 1, out.data_ptr(), s.cuda_stream) == 0
    assert L.adn_perceptual_loss(p.data_ptr(), q.data_ptr(), 1, 40, 64, ws.data_ptr(), need.value, lo.data_ptr(), s.cuda_stream) == 0
    g2.capture_end()
g2.replay(); torch.cuda.synchronize()
assert float(out.abs().max()) > 0 and float(lo[0, 3]) > 0
print("OK")
"""
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_split_k_small_batch_path(dev, weights_np, golden_dir, monkeypatch):
    """ADN_WINO_SPLITK=1 (serving latency at batch 1-2): deep layers are cut along K into up to 8 workgroups per output
    tile + a reduce launch.  Same tolerance against the reference goldens; not bit-identical to the default path
    (different summation order), which is why it is opt-in."""
    from audiodenoiser_amd.model import UNet
    from audiodenoiser_amd.weights import make_input
    monkeypatch.setenv("ADN_WINO_SPLITK", "1")
    m = UNet(1, 1)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in weights_np.items()}, strict=True)
    m = m.to(dev).eval()
    for (n, f, t) in GOLDEN_SHAPES:
        g = np.load(os.path.join(golden_dir, f"unet_{f}x{t}.npz"))
        x = torch.from_numpy(make_input(7, n, f, t)).to(dev)
        with torch.no_grad():
            y, taps = m(x, return_taps=True)
            plain = m(x)
            again = m(x)
        assert _rel(y.cpu().numpy(), g["y"]) <= TOL, (f, t)
        assert _rel(plain.cpu().numpy(), g["y"]) <= TOL, (f, t)
        assert torch.equal(plain, again)                               # fixed summation order: deterministic
        for name, tp in taps.items():
            a = tp.cpu().numpy().astype(np.float64).ravel()
            s_, sa, sq, cnt = g[f"{name}_stats"]
            assert np.abs(a[g[f"{name}_idx"]] - g[f"{name}_val"]).max() <= 10 * TOL * np.sqrt(sq / cnt), name
            assert abs(np.abs(a).sum() - sa) <= TOL * sa, name


def test_fp16_transposed_convolution_kernels_agree(dev, weights_np, monkeypatch):
    """convt16_f16 (16x16x32 MFMA, 16-byte stores through v_permlane16_swap) against conv_dma<_Float16, ..., CONVT2X2>: the same
    fp16 products summed in fp32 in a different order -- the four up-path taps and the output agree to fp16 rounding, on shapes
    with partial tiles (rows / columns beyond the image), several tiles per image and a batch."""
    from audiodenoiser_amd.model import UNet
    from audiodenoiser_amd.weights import make_input

    def net(env):
        for k in ("ADN_F16_CONV", "ADN_F16_FIRST", "ADN_F16_CONVT"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        m = UNet(1, 1)
        m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in weights_np.items()}, strict=True)
        m = m.to(dev).eval().set_compute_dtype("f16")
        with torch.no_grad():
            m(torch.zeros((1, 1, 16, 16), device=dev))
        return m
    new, old = net({}), net({"ADN_F16_CONVT": "dma"})
    monkeypatch.delenv("ADN_F16_CONVT", raising=False)
    for (n, f, t) in ((1, 16, 16), (3, 33, 47), (2, 257, 188), (2, 513, 256), (1, 600, 300), (5, 48, 1040)):
        x = torch.from_numpy(make_input(5, n, f, t)).to(dev)
        with torch.no_grad():
            yn, tn = new(x, return_taps=True)
            yo, to = old(x, return_taps=True)
        for name in ("up1", "up2", "up3", "up4"):
            a, b = tn[name], to[name]
            assert bool(torch.isfinite(a).all())
            d = float((a - b).abs().max()) / float(b.abs().max())
            assert d <= 4e-3, (n, f, t, name, d)                 # (fp16 storage: 1 ulp = 1e-3 relative; later taps compound)
        assert float((yn - yo).abs().max()) <= 4e-3 * float(yo.abs().max()), (n, f, t)
