"""STFT-magnitude oracle: parity UNPINNED at the librosa boundary (librosa absent, no reference fixture).

What can be checked here: the C restatement against the numpy restatement of librosa 0.10 semantics,
against torch.stft (an independent implementation), and against analytic known answers.
"""
import numpy as np
import pytest
import torch

import oracle
from oracle import stft_numpy
from audiodenoiser_amd.weights import make_audio

CASES = [  # (L, n_fft, hop, center)
    (16000, 512, 128, False),   # reference train chunk: create_train_dataset.py:21-27,167-172 -> 257x122
    (24000, 512, 128, True),    # reference test clip: create_test_dataset.py:20-22,39 -> 257x188
    (5000, 1024, 256, True),
    (1024, 1024, 256, False),   # exactly one frame
    (700, 64, 16, True),
]


@pytest.mark.parametrize("L,n_fft,hop,center", CASES)
def test_c_vs_numpy_vs_torch(L, n_fft, hop, center):
    a = make_audio(3, 1, L)[0]
    c = oracle.stft_mag(a, n_fft, hop, center)
    npy = stft_numpy.stft_mag(a, n_fft, hop, center)
    assert c.shape == npy.shape == (n_fft // 2 + 1, stft_numpy.n_frames(L, n_fft, hop, center))
    scale = np.abs(npy).max()
    assert np.abs(c - npy).max() <= 3e-7 * scale
    t = torch.stft(torch.from_numpy(a), n_fft, hop_length=hop, win_length=n_fft,
                   window=torch.hann_window(n_fft, periodic=True), center=center, pad_mode="constant",
                   return_complex=True).abs().numpy()
    assert t.shape == c.shape
    assert np.abs(c - t).max() <= 2e-6 * scale


def test_reference_shapes():
    assert oracle.stft_mag(np.zeros(16000, np.float32), 512, 128, False).shape == (257, 122)
    assert oracle.stft_mag(np.zeros(24000, np.float32), 512, 128, True).shape == (257, 188)
    assert oracle.stft_mag(np.zeros(132300, np.float32), 1024, 256, True).shape == (513, 517)
    assert oracle.stft_mag(np.zeros(132300, np.float32), 1024, 256, False).shape == (513, 513)


def test_known_answers():
    n, hop = 256, 64
    L = n * 4
    i = np.arange(L)
    # on-bin cosine through a periodic Hann window: |X[k0]| = A*N/4, |X[k0+-1]| = A*N/8, 0 elsewhere
    k0, amp = 20, 0.7
    x = (amp * np.cos(2 * np.pi * k0 * i / n)).astype(np.float32)
    m = oracle.stft_mag(x, n, hop, False)
    for f in range(m.shape[1]):
        col = m[:, f]
        assert abs(col[k0] - amp * n / 4) < 1e-3
        assert abs(col[k0 - 1] - amp * n / 8) < 1e-3 and abs(col[k0 + 1] - amp * n / 8) < 1e-3
        rest = np.delete(col, [k0 - 1, k0, k0 + 1])
        assert rest.max() < 1e-3
    # DC: |X[0]| = N/2, |X[1]| = N/4
    m = oracle.stft_mag(np.ones(L, np.float32), n, hop, False)
    assert np.allclose(m[0], n / 2, atol=1e-3) and np.allclose(m[1], n / 4, atol=1e-3)
    assert m[2:].max() < 1e-3
    # unit impulse at sample s of frame 0 -> flat spectrum equal to the window value at s
    x = np.zeros(L, np.float32)
    s = 77
    x[s] = 1.0
    m = oracle.stft_mag(x, n, hop, False)
    w = 0.5 - 0.5 * np.cos(2 * np.pi * s / n)
    assert np.allclose(m[:, 0], w, atol=1e-6)
    # centre padding is zeros (librosa 0.10 default pad_mode="constant"): first frame sees n/2 zeros
    m = oracle.stft_mag(np.ones(L, np.float32), n, hop, True)
    assert abs(m[0, 0] - (n / 4 + 0.5)) < 1e-3   # second half of the window: w[n/2]=1 plus (n/2-1)/2
    assert abs(m[0, 2] - n / 2) < 1e-3      # frame 2 starts at sample 0: full window


def test_too_short_raises():
    with pytest.raises(ValueError):
        oracle.stft_mag(np.zeros(100, np.float32), 512, 128, False)
