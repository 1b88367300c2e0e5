"""The CPU oracle against golden vectors produced by the reference's own model.py / data_loader.py.

Fixtures: tests/golden/*.npz, written by tools/make_golden.py (which imports /root/reference/code).
These tests pin the oracle; the -m gpu tests then compare the HIP path with the oracle and the goldens.
"""
import os

import numpy as np
import pytest

from conftest import VARIANT_KINDS, VARIANT_SHAPES, load_real_audio_fixture, load_variant_golden, variant_input
import torch

import oracle
from oracle import unet_torch
from audiodenoiser_amd.weights import make_input

TAPS = oracle.TAP_NAMES
SMALL = ((2, 16, 16), (2, 33, 47), (1, 64, 80))
ALL = SMALL + ((1, 257, 188), (1, 513, 256))


def _load(golden_dir, f, t):
    g = np.load(os.path.join(golden_dir, f"unet_{f}x{t}.npz"))
    assert int(g["weight_seed"]) == 1234 and int(g["input_seed"]) == 7
    return g


def _check_taps(g, taps, tol):
    for name in TAPS:
        a = np.asarray(taps[name], dtype=np.float64).ravel()
        s, sa, sq, cnt = g[f"{name}_stats"]
        assert a.size == int(cnt), name
        scale = np.sqrt(sq / cnt)
        got = a[g[f"{name}_idx"]]
        assert np.abs(got - g[f"{name}_val"]).max() <= tol * max(scale, 1e-6) * 10, name
        assert abs(np.abs(a).sum() - sa) <= tol * sa, name
        assert abs((a * a).sum() - sq) <= 2 * tol * sq, name


@pytest.mark.parametrize("n,f,t", SMALL)
@pytest.mark.parametrize("acc64", [False, True])
def test_c_oracle_matches_reference_golden(golden_dir, weights_np, n, f, t, acc64):
    g = _load(golden_dir, f, t)
    x = make_input(7, n, f, t)
    y, taps = oracle.unet_forward(weights_np, x, acc64=acc64, want_taps=True)
    ref = g["y"]
    assert y.shape == ref.shape
    # relative to max|y|: fp32 re-association noise of an 18-conv-deep net is ~2e-6 (measured).
    assert np.abs(y - ref).max() <= 1e-5 * np.abs(ref).max()
    _check_taps(g, taps, 1e-5)


def test_c_oracle_reference_test_shape(golden_dir, weights_np):
    """257x188 = the reference's own test-set spectrogram shape (create_test_dataset.py:39); pads in W at every level."""
    g = _load(golden_dir, 257, 188)
    y = oracle.unet_forward(weights_np, make_input(7, 1, 257, 188))
    assert np.abs(y - g["y"]).max() <= 1e-5 * np.abs(g["y"]).max()


@pytest.mark.parametrize("n,f,t", ALL)
def test_torch_oracle_matches_reference_golden(golden_dir, weights_np, n, f, t):
    g = _load(golden_dir, f, t)
    sd = unet_torch.to_torch_state(weights_np)
    y, taps = unet_torch.unet_forward(sd, torch.from_numpy(make_input(7, n, f, t)), want_taps=True)
    ref = g["y"]
    assert np.abs(y.numpy() - ref).max() <= 1e-6 * np.abs(ref).max()
    _check_taps(g, {k: v.numpy() for k, v in taps.items()}, 1e-6)


def test_oracle_rejects_too_small():
    with pytest.raises(ValueError):
        oracle.unet_forward({f"k{i}": np.zeros(1, np.float32) for i in range(118)}, np.zeros((1, 1, 8, 8), np.float32))


def test_loader_quantize_pad_matches_reference(golden_dir):
    """fp16 round trip (overflow->inf, underflow->0) + crop / bottom-right zero pad, data_loader.py:41-72."""
    from audiodenoiser_amd.weights import hash_uniform
    g = np.load(os.path.join(golden_dir, "loader_cases.npz"))
    for ci in range(4):
        shape = tuple(g[f"case{ci}_in_shape"])
        target = tuple(g[f"case{ci}_target"])
        u = hash_uniform(5, f"loader{ci}", 2 * shape[0] * shape[1]).reshape(2, *shape)
        noisy = (u[0] * np.float32(8.0)).astype(np.float32)
        clean = (u[1] * np.float32(8.0)).astype(np.float32)
        noisy[0, 0], noisy[0, 1], noisy[0, 2], noisy[1, 0] = 70000.0, 1e-8, 3e-6, 65504.0
        for src, key in ((noisy, "noisy"), (clean, "clean")):
            got = oracle.quantize_pad(src, target)
            ref = g[f"case{ci}_{key}"]
            assert ref.shape == (1,) + target
            assert np.array_equal(got, ref[0]), (ci, key)
        assert np.isinf(g[f"case{ci}_noisy"][0, 0, 0]) and g[f"case{ci}_noisy"][0, 0, 1] == 0.0


def _real_clip(golden_dir):
    fx = load_real_audio_fixture(golden_dir)
    assert int(fx["sample_rate"]) == 44100 and fx["lr_sum_int16"].shape == (132300,)
    return fx["lr_sum_int16"].astype(np.float32) / np.float32(65536.0)      # mean(L, R) / 32768, exact in fp32


def test_config0_real_audio_oracle_chain_matches_reference_golden(golden_dir, weights_np):
    """BASELINE configs[0] on the bundled real clip (tools/make_real_audio_fixture.py): oracle STFT 1024/256 centred
    -> loader rule -> forward.  The loader and forward stages are pinned by config0_real_audio.npz, which the
    reference's own data_loader.py + model.py produced from the same STFT (tools/make_golden.py --only config0); the
    STFT stage is parity unpinned (librosa absent) and cross-checked in tests/test_stft_oracle.py."""
    g = load_real_audio_fixture(golden_dir, "config0_real_audio.npz")
    clip = _real_clip(golden_dir)
    assert 0.1 < np.abs(clip).max() < 0.2 and clip.std() < 0.02            # quiet real recording, 21 dB crest factor
    mag = oracle.stft_mag(clip, 1024, 256, True)
    assert mag.shape == (513, 517)
    x = oracle.quantize_pad(mag, (513, 256))
    assert np.array_equal(x, g["x_f16"].astype(np.float32))                # loader rule, bit exact
    sd = unet_torch.to_torch_state(weights_np)
    y = unet_torch.unet_forward(sd, torch.from_numpy(x[None, None])).numpy()[0, 0]
    assert np.abs(y - g["y"]).max() <= 1e-6 * np.abs(g["y"]).max()


# ---- goldens under parameter distributions the benign set never shows (tools/make_golden.py --only variants) ----
@pytest.mark.parametrize("kind", VARIANT_KINDS)
@pytest.mark.parametrize("f,t", VARIANT_SHAPES)
def test_torch_oracle_matches_variant_goldens(golden_dir, variant_weights, kind, f, t):
    """Trained-like BatchNorm statistics (running_var over five decades, negative gammas) and heavy-tailed weights, real-audio
    magnitudes at scale 1 and 100: the reference's own forward (model.py:53-94) froze these; the torch restatement must
    reproduce them like the benign ones."""
    g = load_variant_golden(golden_dir, kind, f, t)
    sd = unet_torch.to_torch_state(variant_weights(kind))
    y, taps = unet_torch.unet_forward(sd, torch.from_numpy(variant_input(golden_dir, f, t)), want_taps=True)
    ref = g["y"]
    assert ref.shape == (2, 1, f, t) and np.isfinite(ref).all()
    for clip in range(2):                                       # per clip: the x100 clip must not mask the other
        assert np.abs(y.numpy()[clip] - ref[clip]).max() <= 1e-6 * np.abs(ref[clip]).max(), clip
    _check_taps(g, {k: v.numpy() for k, v in taps.items()}, 1e-6)


@pytest.mark.parametrize("kind", VARIANT_KINDS)
def test_c_oracle_matches_variant_golden_small(golden_dir, variant_weights, kind):
    g = load_variant_golden(golden_dir, kind, 33, 47)
    y, taps = oracle.unet_forward(variant_weights(kind), variant_input(golden_dir, 33, 47), acc64=True, want_taps=True)
    for clip in range(2):
        assert np.abs(y[clip] - g["y"][clip]).max() <= 1e-5 * np.abs(g["y"][clip]).max(), clip
    _check_taps(g, taps, 1e-5)


def test_weight_variants_are_what_they_claim(variant_weights):
    tr, hv = variant_weights("trained"), variant_weights("heavy")
    var = np.concatenate([v.ravel() for k, v in tr.items() if k.endswith("running_var")])
    gam = np.concatenate([v.ravel() for k, v in tr.items() if ".double_conv." in k and k.endswith((".1.weight", ".4.weight"))])
    assert 9e-4 < var.min() < 2e-3 and 50 < var.max() <= 105          # five decades
    assert gam.min() < -1.4 and gam.max() > 1.4 and np.abs(gam).min() < 1e-3
    w = hv["bottleneck.double_conv.3.weight"].ravel()
    bound = np.sqrt(6.0 / (1024 * 9))
    assert (np.abs(w) > 5 * bound).sum() in range(1, 9)               # the 8 outliers (fewer if two hashes collide / tiny values)


def test_torch_oracle_matches_reference_golden_two_planes_three_classes(golden_dir):
    """UNet(in_channels=2, num_classes=3) (model.py:54,56,68; tools/make_golden.py --only channels)."""
    from audiodenoiser_amd.weights import make_state_dict
    g = np.load(os.path.join(golden_dir, "unet_c2k3_33x47.npz"))
    sd = unet_torch.to_torch_state(make_state_dict(1234, 2, 3))
    x = torch.from_numpy(make_input(7, 4, 33, 47).reshape(2, 2, 33, 47))
    y, taps = unet_torch.unet_forward(sd, x, want_taps=True)
    assert tuple(y.shape) == (2, 3, 33, 47)
    assert np.abs(y.numpy() - g["y"]).max() <= 1e-6 * np.abs(g["y"]).max()
    _check_taps(g, {k: v.numpy() for k, v in taps.items()}, 1e-6)


NONFINITE_POS = {(257, 188): (20, 20), (1100, 48): (40, 20)}


def _nonfinite_input(f, t, kind):
    x = make_input(7, 1, f, t).copy()
    r, c = NONFINITE_POS[(f, t)]
    x[0, 0, r, c] = np.float32(np.inf) if kind == "inf" else np.float32(np.nan)
    return x


@pytest.mark.parametrize("f,t", list(NONFINITE_POS))
@pytest.mark.parametrize("kind", ["inf", "nan"])
def test_oracles_carry_nonfinite_pixels_as_the_reference_does(golden_dir, weights_np, f, t, kind):
    """unet_nonfinite_<F>x<T>.npz (tools/make_golden.py --only nonfinite: the reference's forward on an input with one +inf /
    NaN pixel; nn.ReLU and nn.MaxPool2d propagate NaN, model.py:13,16,26): both oracles reproduce the non-finite set exactly
    and the finite values outside it."""
    g = np.load(os.path.join(golden_dir, f"unet_nonfinite_{f}x{t}.npz"))
    ref_bad = np.unpackbits(g[f"{kind}_mask"])[: f * t].reshape(f, t).astype(bool)
    assert ref_bad.any() and not ref_bad.all()
    x = _nonfinite_input(f, t, kind)
    ys = [unet_torch.unet_forward(unet_torch.to_torch_state(weights_np), torch.from_numpy(x)).numpy()[0, 0]]
    if (f, t) == (1100, 48):                                 # the C restatement on the cheaper shape only
        ys.append(oracle.unet_forward(weights_np, x)[0, 0])
    for y in ys:
        bad = ~np.isfinite(y)
        assert np.array_equal(bad, ref_bad)
        assert np.abs(y[~bad] - g[f"{kind}_y"][~bad]).max() <= 1e-5 * np.abs(g[f"{kind}_y"]).max()
