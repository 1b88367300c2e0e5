#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own code in the build container.

Run once here (``python tools/make_golden.py``); the reference (``/root/reference``) does not exist on
the GPU box, so only the small outputs are committed.  Inputs and weights are NOT stored: they are
regenerated bit-identically from ``audiodenoiser_amd.weights`` (hash PRNG) by the tests.

What is frozen:

* ``unet_<F>x<T>.npz`` — ``model.UNet(1,1).eval()`` (reference ``code/model.py:53-94``) loaded (strict) with
  ``make_state_dict(seed)``, forward under ``torch.no_grad`` on ``make_input(seed_x, N, F, T)``:
  the full output, and per block output (down1..4, bottleneck, up1..4, out) float64 statistics
  (sum, sum|.|, sum of squares) plus 512 sampled elements at hash-chosen flat indices.
* ``loader_cases.npz`` — ``data_loader.SpectrogramDataset`` (reference ``code/data_loader.py:37-72``) run on
  temporary .npy files: fp16 round trip + crop/pad for four (shape, target_size) cases.
* ``config0_real_audio.npz`` (``--only config0``) — BASELINE configs[0] on REAL audio: the 3 s clip frozen by
  ``tools/make_real_audio_fixture.py`` -> STFT 1024/256 centred (CPU oracle; librosa is absent, that stage is parity
  unpinned) saved as .npy -> the reference's ``SpectrogramDataset(target_size=(513, 256))`` -> the reference's
  ``UNet.forward``.  Stored: ``x_f16`` (the loader's output, exactly representable in fp16) and ``y``.
* ``unet_{trained,heavy}_<F>x<T>.npz`` (``--only variants``) — the reference forward under two more seeded parameter sets
  (``audiodenoiser_amd.weights.make_state_dict_variant``: trained-like BatchNorm statistics / heavy-tailed weights) on the
  real-audio input above cropped to (F, T), two clips at scale 1 and 100.
* ``unet_c2k3_33x47.npz`` (``--only channels``) — the reference's ``UNet(in_channels=2, num_classes=3)`` on a seeded input.
* ``unet_nonfinite_<F>x<T>.npz`` (``--only nonfinite``) — the reference forward on a seeded input with one +inf / one NaN pixel
  (what ``data_loader.py:41-42`` produces for a magnitude above 65504): non-finite masks + the finite values outside them.
* ``loss_cases.npz`` (``--only loss``) — the reference's ``MultiScaleSTFTLoss`` (``code/loss.py:6-35``) and ``nn.L1Loss``
  (``loss.py:75,86``) on three seeded (B, 1, F, T) pairs.  The mel term needs torchaudio (absent): not frozen, parity unpinned.
"""
from __future__ import annotations

import contextlib
import io
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/code")

from audiodenoiser_amd.weights import hash_uniform, make_input, make_state_dict, make_state_dict_variant  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
WEIGHT_SEED = 1234
INPUT_SEED = 7
UNET_CASES = ((2, 16, 16), (2, 33, 47), (1, 64, 80), (1, 257, 188), (1, 513, 256))
N_SAMPLES = 512
TAPS = ("down1", "down2", "down3", "down4", "bottleneck", "upconv1", "upconv2", "upconv3", "upconv4", "out")
TAP_KEYS = ("down1", "down2", "down3", "down4", "bottleneck", "up1", "up2", "up3", "up4", "out")


def sample_indices(name: str, numel: int) -> np.ndarray:
    u = hash_uniform(99, "sample:" + name, N_SAMPLES).astype(np.float64)
    return np.minimum((u * numel).astype(np.int64), numel - 1)


def config0_real_audio() -> None:
    import data_loader as ref_loader  # reference code/data_loader.py
    import model as ref_model  # reference code/model.py
    import oracle

    fx = np.load(os.path.join(GOLDEN, "real_audio_17480-2-0-24.npz"))
    clip = fx["lr_sum_int16"].astype(np.float32) / np.float32(65536.0)          # mono mix, exact
    mag = oracle.stft_mag(clip, 1024, 256, True)                                 # (513, 517)
    torch.set_num_threads(os.cpu_count() or 1)
    net = ref_model.UNet(in_channels=1, num_classes=1)
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in make_state_dict(WEIGHT_SEED).items()}, strict=True)
    net.eval()
    with tempfile.TemporaryDirectory() as d:
        np.save(os.path.join(d, "noisy_real_chunk_0.npy"), mag)
        np.save(os.path.join(d, "clean_real_chunk_0.npy"), mag)
        with contextlib.redirect_stdout(io.StringIO()):
            ds = ref_loader.SpectrogramDataset(d, target_size=(513, 256))
        x, _ = ds[0]                                                             # (1, 513, 256) fp32(fp16(.))
    assert np.array_equal(x.numpy().astype(np.float16).astype(np.float32), x.numpy())
    with torch.no_grad():
        y = net(x[None])
    path = os.path.join(GOLDEN, "config0_real_audio.npz")
    np.savez_compressed(path, x_f16=x.numpy()[0].astype(np.float16), y=y.numpy()[0, 0].astype(np.float32),
                        weight_seed=np.array(WEIGHT_SEED))
    print(f"wrote {path} ({os.path.getsize(path)} bytes): x max {float(x.max()):.3f}, y mean {float(y.mean()):+.4f} "
          f"std {float(y.std()):.4f}")


VARIANT_KINDS = ("trained", "heavy")
VARIANT_SHAPES = ((33, 47), (257, 188), (513, 256))
REF_MODULES = ("downconv1", "downconv2", "downconv3", "downconv4", "bottleneck", "upconv1", "upconv2", "upconv3", "upconv4", "out")


def variant_input(golden_dir: str, f: int, t: int) -> np.ndarray:
    """Input of the weight-variant goldens: the real-audio network input of configs[0] (``config0_real_audio.npz``: the
    reference loader's fp16-rounded 513x256 magnitude) cropped to (f, t) by the loader rule (top-left), as two clips:
    clip 0 at scale 1, clip 1 at scale 100 (fp32 product)."""
    x16 = np.load(os.path.join(golden_dir, "config0_real_audio.npz"))["x_f16"]
    x = x16[:f, :t].astype(np.float32)
    return np.stack([x, x * np.float32(100.0)])[:, None]


def weight_variants() -> None:
    """``unet_<kind>_<F>x<T>.npz``: the reference's ``UNet.forward`` (code/model.py:53-94) under parameter distributions the
    benign goldens never show (``make_state_dict_variant``: trained-like BatchNorm statistics, heavy-tailed weights) on
    real-audio magnitudes at scale 1 and 100.  Same record layout as ``unet_<F>x<T>.npz``."""
    import model as ref_model  # reference code/model.py

    torch.set_num_threads(os.cpu_count() or 1)
    for kind in VARIANT_KINDS:
        net = ref_model.UNet(in_channels=1, num_classes=1)
        sd = make_state_dict_variant(kind, WEIGHT_SEED)
        net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, strict=True)
        net.eval()
        acts = {}

        def hook(name):
            def fn(_mod, _inp, out):
                acts[name] = (out[0] if isinstance(out, tuple) else out).detach()
            return fn

        for mod_name in REF_MODULES:
            getattr(net, mod_name).register_forward_hook(hook(mod_name))
        for f, t in VARIANT_SHAPES:
            x = torch.from_numpy(variant_input(GOLDEN, f, t))
            with torch.no_grad():
                y = net(x)
            assert bool(torch.isfinite(y).all()), (kind, f, t)
            rec = {"y": y.numpy().astype(np.float32), "shape": np.array([2, f, t]), "weight_seed": np.array(WEIGHT_SEED)}
            line = []
            for mod_name, key in zip(REF_MODULES, TAP_KEYS):
                a = acts[mod_name].numpy().astype(np.float64).ravel()
                idx = sample_indices(key, a.size)
                rec[f"{key}_stats"] = np.array([a.sum(), np.abs(a).sum(), (a * a).sum(), a.size], dtype=np.float64)
                rec[f"{key}_idx"] = idx
                rec[f"{key}_val"] = a[idx].astype(np.float32)
                line.append(f"{key} {np.abs(a).max():.3g}")
            path = os.path.join(GOLDEN, f"unet_{kind}_{f}x{t}.npz")
            np.savez_compressed(path, **rec)
            print(f"wrote {path} ({os.path.getsize(path)} B): y mean {float(y.mean()):+.4g} std {float(y.std()):.4g}; max|tap|: " + ", ".join(line))


def channels_case() -> None:
    """``unet_c2k3_33x47.npz``: the reference's ``UNet(in_channels=2, num_classes=3)`` (code/model.py:54,56,68 accept any; its own
    callers use (1, 1)) on a seeded (2, 2, 33, 47) input: full output (2, 3, 33, 47) + the block statistics of the other goldens."""
    import model as ref_model  # reference code/model.py

    torch.set_num_threads(os.cpu_count() or 1)
    net = ref_model.UNet(in_channels=2, num_classes=3)
    sd = make_state_dict(WEIGHT_SEED, 2, 3)
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, strict=True)
    net.eval()
    acts = {}

    def hook(name):
        def fn(_mod, _inp, out):
            acts[name] = (out[0] if isinstance(out, tuple) else out).detach()
        return fn

    for mod_name in REF_MODULES:
        getattr(net, mod_name).register_forward_hook(hook(mod_name))
    n, f, t = 2, 33, 47
    x = torch.from_numpy(make_input(INPUT_SEED, n * 2, f, t).reshape(n, 2, f, t))
    with torch.no_grad():
        y = net(x)
    rec = {"y": y.numpy().astype(np.float32), "shape": np.array([n, 2, 3, f, t]), "weight_seed": np.array(WEIGHT_SEED),
           "input_seed": np.array(INPUT_SEED)}
    for mod_name, key in zip(REF_MODULES, TAP_KEYS):
        a = acts[mod_name].numpy().astype(np.float64).ravel()
        idx = sample_indices(key, a.size)
        rec[f"{key}_stats"] = np.array([a.sum(), np.abs(a).sum(), (a * a).sum(), a.size], dtype=np.float64)
        rec[f"{key}_idx"] = idx
        rec[f"{key}_val"] = a[idx].astype(np.float32)
    path = os.path.join(GOLDEN, "unet_c2k3_33x47.npz")
    np.savez_compressed(path, **rec)
    print(f"wrote {path}: y{tuple(y.shape)} mean {float(y.mean()):+.4f} std {float(y.std()):.4f}")


# ---- non-finite inputs -------------------------------------------------------------------------------------------------
# (F, T) -> (row, column) of the poisoned pixel: near the top-left corner, so that the rows beyond the network's receptive
# field (185 pixels; more for the Winograd kernels, whose tiles round the poisoned set outward) stay finite and are compared.
# 1100x48: a tall image for the F(4x4,3x3) kernel, whose bound on the poisoned set exceeds 257 / 513 rows.
NONFINITE_CASES = {(257, 188): (20, 20), (513, 256): (60, 40), (1100, 48): (40, 20)}
NONFINITE_KINDS = {"inf": np.float32(np.inf), "nan": np.float32(np.nan)}


def nonfinite_input(f: int, t: int, kind: str) -> np.ndarray:
    """Input of the non-finite goldens: ``make_input(INPUT_SEED, 1, f, t)`` with ONE pixel replaced by +inf / NaN -- what the
    reference's own loader hands the network for a magnitude above 65504 (``data_loader.py:41-42``; loader_cases.npz [0, 0])."""
    x = make_input(INPUT_SEED, 1, f, t).copy()
    r, c = NONFINITE_CASES[(f, t)]
    x[0, 0, r, c] = NONFINITE_KINDS[kind]
    return x


def nonfinite_cases() -> None:
    """``unet_nonfinite_<F>x<T>.npz`` (``--only nonfinite``): the reference's ``UNet.forward`` (code/model.py:53-94; nn.ReLU :13,16
    and nn.MaxPool2d :26 propagate NaN) on a seeded input with one +inf / one NaN pixel.  Stored per kind: the output's non-finite
    mask (packed bits), the output with the non-finite elements set to 0, and per block output the packed mask of the pixels at
    which ANY channel is non-finite."""
    import model as ref_model  # reference code/model.py

    torch.set_num_threads(os.cpu_count() or 1)
    net = ref_model.UNet(in_channels=1, num_classes=1)
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in make_state_dict(WEIGHT_SEED).items()}, strict=True)
    net.eval()
    acts = {}

    def hook(name):
        def fn(_mod, _inp, out):
            acts[name] = (out[0] if isinstance(out, tuple) else out).detach()
        return fn

    for mod_name in REF_MODULES:
        getattr(net, mod_name).register_forward_hook(hook(mod_name))
    for (f, t), pos in NONFINITE_CASES.items():
        rec = {"shape": np.array([1, f, t]), "pos": np.array(pos), "weight_seed": np.array(WEIGHT_SEED), "input_seed": np.array(INPUT_SEED)}
        for kind in NONFINITE_KINDS:
            with torch.no_grad():
                y = net(torch.from_numpy(nonfinite_input(f, t, kind))).numpy()[0, 0]
            bad = ~np.isfinite(y)
            rec[f"{kind}_mask"] = np.packbits(bad)
            rec[f"{kind}_y"] = np.where(bad, np.float32(0), y).astype(np.float32)
            line = []
            for mod_name, key in zip(REF_MODULES, TAP_KEYS):
                a = acts[mod_name].numpy()[0]                                   # (C, h, w)
                pm = (~np.isfinite(a)).any(axis=0)
                rec[f"{kind}_{key}_mask"] = np.packbits(pm)
                rec[f"{kind}_{key}_hw"] = np.array(pm.shape)
                line.append(f"{key} {int(pm.sum())}/{pm.size}")
            rows = np.flatnonzero(bad.any(axis=1))
            print(f"nonfinite {f}x{t} {kind} at {pos}: {int(bad.sum())} of {bad.size} outputs non-finite (rows {rows.min()}..{rows.max()}), "
                  f"{int(np.isnan(y).sum())} NaN; poisoned pixels per block: " + ", ".join(line))
        path = os.path.join(GOLDEN, f"unet_nonfinite_{f}x{t}.npz")
        np.savez_compressed(path, **rec)
        print(f"wrote {path} ({os.path.getsize(path)} B)")


LOSS_CASES = ((3, 40, 96), (2, 257, 188), (2, 513, 256))


def loss_inputs(case: int, b: int, f: int, t: int):
    """(pred, target) of loss golden `case`: hash-PRNG magnitudes, target = a different stream (losses of order 1)."""
    pred = hash_uniform(21, f"loss_pred{case}", b * f * t).reshape(b, 1, f, t) * np.float32(3.0)
    target = hash_uniform(22, f"loss_target{case}", b * f * t).reshape(b, 1, f, t) * np.float32(3.0)
    return pred.astype(np.float32), target.astype(np.float32)


def loss_cases() -> None:
    """``loss_cases.npz``: the torch-only terms of the reference's ``CombinedPerceptualLoss`` (code/loss.py:6-35,86) from the
    reference's own code.  ``loss.py:4`` imports torchaudio (absent here) at module level; an EMPTY placeholder module is
    registered under that name solely so that ``import loss`` resolves.  Only ``loss.MultiScaleSTFTLoss()`` and
    ``torch.nn.L1Loss()`` are instantiated — neither touches torchaudio.  ``MelSpectrogramLoss`` /
    ``CombinedPerceptualLoss`` are NOT instantiated (they need the real library): the mel term stays parity unpinned."""
    import types
    if "torchaudio" not in sys.modules:
        sys.modules["torchaudio"] = types.ModuleType("torchaudio")        # empty: any attribute access raises
    import loss as ref_loss  # reference code/loss.py

    stft_loss = ref_loss.MultiScaleSTFTLoss()
    l1_loss = torch.nn.L1Loss()                                           # what CombinedPerceptualLoss.__init__ holds (loss.py:75)
    rec = {"cases": np.array(LOSS_CASES)}
    for ci, (b, f, t) in enumerate(LOSS_CASES):
        pred, target = (torch.from_numpy(a) for a in loss_inputs(ci, b, f, t))
        with torch.no_grad():
            s = stft_loss(pred, target)
            l1 = l1_loss(pred, target)
        rec[f"case{ci}_stft"] = np.array(float(s), dtype=np.float64)
        rec[f"case{ci}_l1"] = np.array(float(l1), dtype=np.float64)
        print(f"loss case {ci} {(b, f, t)}: stft {float(s):.7f} l1 {float(l1):.7f}")
    path = os.path.join(GOLDEN, "loss_cases.npz")
    np.savez_compressed(path, **rec)
    print("wrote", path)


def main() -> None:
    if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "config0":
        return config0_real_audio()
    if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "loss":
        return loss_cases()
    if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "channels":
        return channels_case()
    if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "variants":
        return weight_variants()
    if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "nonfinite":
        return nonfinite_cases()
    import data_loader as ref_loader  # reference code/data_loader.py
    import model as ref_model  # reference code/model.py

    os.makedirs(GOLDEN, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 1)
    net = ref_model.UNet(in_channels=1, num_classes=1)
    sd = make_state_dict(WEIGHT_SEED)
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, strict=True)
    net.eval()

    acts = {}

    def hook(name):
        def f(_mod, _inp, out):
            acts[name] = (out[0] if isinstance(out, tuple) else out).detach()
        return f

    for mod_name in ("downconv1", "downconv2", "downconv3", "downconv4", "bottleneck",
                     "upconv1", "upconv2", "upconv3", "upconv4", "out"):
        getattr(net, mod_name).register_forward_hook(hook(mod_name))

    for n, f, t in UNET_CASES:
        x = torch.from_numpy(make_input(INPUT_SEED, n, f, t))
        with torch.no_grad():
            y = net(x)
        rec = {"y": y.numpy().astype(np.float32), "shape": np.array([n, f, t]),
               "weight_seed": np.array(WEIGHT_SEED), "input_seed": np.array(INPUT_SEED)}
        for mod_name, key in zip(("downconv1", "downconv2", "downconv3", "downconv4", "bottleneck",
                                  "upconv1", "upconv2", "upconv3", "upconv4", "out"), TAP_KEYS):
            a = acts[mod_name].numpy().astype(np.float64).ravel()
            idx = sample_indices(key, a.size)
            rec[f"{key}_stats"] = np.array([a.sum(), np.abs(a).sum(), (a * a).sum(), a.size], dtype=np.float64)
            rec[f"{key}_idx"] = idx
            rec[f"{key}_val"] = a[idx].astype(np.float32)
        path = os.path.join(GOLDEN, f"unet_{f}x{t}.npz")
        np.savez_compressed(path, **rec)
        print(f"wrote {path}: y{tuple(y.shape)} mean {float(y.mean()):+.4f} std {float(y.std()):.4f}")

    # ---- loader cases (fp16 quantise + crop/pad), reference data_loader.SpectrogramDataset ----
    cases = (((257, 122), (256, 64)), ((20, 30), (32, 40)), ((40, 30), (32, 40)), ((257, 188), (513, 256)))
    rec = {}
    for ci, (shape, target) in enumerate(cases):
        u = hash_uniform(5, f"loader{ci}", 2 * shape[0] * shape[1]).reshape(2, *shape)
        noisy = (u[0] * np.float32(8.0)).astype(np.float32)
        clean = (u[1] * np.float32(8.0)).astype(np.float32)
        # exercise fp16 overflow (> 65504 -> inf), underflow (-> 0) and subnormals
        noisy[0, 0], noisy[0, 1], noisy[0, 2], noisy[1, 0] = 70000.0, 1e-8, 3e-6, 65504.0
        with tempfile.TemporaryDirectory() as d:
            np.save(os.path.join(d, "noisy_a_chunk_0.npy"), noisy)
            np.save(os.path.join(d, "clean_a_chunk_0.npy"), clean)
            with contextlib.redirect_stdout(io.StringIO()):
                ds = ref_loader.SpectrogramDataset(d, target_size=target)
            with np.errstate(over="ignore"):
                n_t, c_t = ds[0]
        rec[f"case{ci}_in_shape"] = np.array(shape)
        rec[f"case{ci}_target"] = np.array(target)
        rec[f"case{ci}_noisy"] = n_t.numpy()
        rec[f"case{ci}_clean"] = c_t.numpy()
    path = os.path.join(GOLDEN, "loader_cases.npz")
    np.savez_compressed(path, **rec)
    print("wrote", path)
    config0_real_audio()
    weight_variants()
    loss_cases()
    channels_case()
    nonfinite_cases()


if __name__ == "__main__":
    main()
