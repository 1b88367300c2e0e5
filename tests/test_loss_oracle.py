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
