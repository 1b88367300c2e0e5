"""CombinedPerceptualLoss oracle (oracle/loss_torch.py) — the reference's loss.py needs torchaudio (absent), so
the mel term is PARITY UNPINNED; the STFT and L1 terms are the reference's own torch calls."""
import numpy as np
import torch

from oracle import loss_torch as L


def test_per_clip_means_equal_reference_batch_formulas():
    g = torch.Generator().manual_seed(0)
    a = torch.rand((3, 1, 40, 96), generator=g) * 3
    b = torch.rand((3, 1, 40, 96), generator=g) * 3
    pc = L.per_clip(a, b)
    stft, l1 = L.batch_reference_parts(a, b)          # written like loss.py:12-35 and :86 (batch l1_loss)
    assert abs(float(pc[:, 1].mean()) - float(stft)) < 1e-6
    assert abs(float(pc[:, 3].mean()) - float(l1)) < 1e-6
    assert torch.allclose(pc[:, 0], 0.4 * pc[:, 1] + 0.4 * pc[:, 2] + 0.2 * pc[:, 3], atol=1e-6)


def test_mel_filterbank_shape_and_known_properties():
    fb = L.mel_filterbank()
    assert fb.shape == (32, 64) and fb.min() >= 0 and fb.max() <= 1.0
    # 64 HTK filters over 32 linear bins up to 4 kHz: the narrow low-frequency triangles fall between bins
    assert 10 <= int((fb.sum(axis=0) == 0).sum()) <= 20
    # centre frequencies rise monotonically where defined
    peaks = [int(np.argmax(fb[:, m])) for m in range(64) if fb[:, m].sum() > 0]
    assert peaks == sorted(peaks)


def test_identical_inputs_give_zero_loss():
    x = torch.rand((2, 1, 33, 80), generator=torch.Generator().manual_seed(1))
    assert float(L.per_clip(x, x.clone()).abs().max()) == 0.0


def test_torch_only_terms_match_the_references_own_loss_py(golden_dir):
    """tests/golden/loss_cases.npz: the reference's own ``MultiScaleSTFTLoss`` (loss.py:6-35) and ``nn.L1Loss`` (loss.py:75,86)
    on seeded pairs (tools/make_golden.py --only loss; torchaudio absent there, so only these two terms were instantiated).
    The oracle's per-clip columns 1 (stft) and 3 (l1) must average to them; the mel column stays parity unpinned."""
    import os
    from audiodenoiser_amd.weights import hash_uniform
    g = np.load(os.path.join(golden_dir, "loss_cases.npz"))
    for ci, (b, f, t) in enumerate(g["cases"]):
        pred = torch.from_numpy(hash_uniform(21, f"loss_pred{ci}", b * f * t).reshape(b, 1, f, t) * np.float32(3.0))
        target = torch.from_numpy(hash_uniform(22, f"loss_target{ci}", b * f * t).reshape(b, 1, f, t) * np.float32(3.0))
        pc = L.per_clip(pred, target).double()
        assert abs(float(pc[:, 1].mean()) - float(g[f"case{ci}_stft"])) <= 2e-6 * float(g[f"case{ci}_stft"]), ci
        assert abs(float(pc[:, 3].mean()) - float(g[f"case{ci}_l1"])) <= 2e-6 * float(g[f"case{ci}_l1"]), ci
        stft, l1 = L.batch_reference_parts(pred, target)
        assert abs(float(stft) - float(g[f"case{ci}_stft"])) <= 1e-6 and abs(float(l1) - float(g[f"case{ci}_l1"])) <= 1e-6
