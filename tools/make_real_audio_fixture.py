#!/usr/bin/env python3
"""Freeze one bundled reference wav as a small real-audio fixture for BASELINE configs[0] (SURVEY.md §0 D4, §8d
Config 1: the IRMAS clips are not shipped, a bundled noise wav stands in).

Runs in the BUILD container only (reads /root/reference as DATA through this repo's own RIFF reader; nothing of
the reference's code is imported or copied):

    python tools/make_real_audio_fixture.py

Writes tests/golden/real_audio_17480-2-0-24.npz:
    lr_sum_int16  (132300,) int16  = left + right of the first 3 s (132 300 frames @ 44.1 kHz) of
                                     /root/reference/data/test/noise/17480-2-0-24.wav (int16 stereo, 4.00 s)
    sample_rate   44100
The mono mix librosa.load(..., mono=True) would produce is mean(L, R) / 32768 = lr_sum_int16 / 65536, exact in fp32
(|L + R| <= 10580 for this file, so the sum fits int16).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SRC = "/root/reference/data/test/noise/17480-2-0-24.wav"
OUT = os.path.join(ROOT, "tests", "golden", "real_audio_17480-2-0-24.npz")
N = 132300


def main():
    from audiodenoiser_amd.wav import read_wav
    audio, rate = read_wav(SRC, mono=False)
    assert rate == 44100 and audio.ndim == 2 and audio.shape[1] == 2 and audio.shape[0] >= N, (rate, audio.shape)
    pcm = np.round(audio[:N].astype(np.float64) * 32768.0).astype(np.int64)
    assert np.array_equal(pcm / 32768.0, audio[:N].astype(np.float64)), "source is not int16 PCM"
    s = pcm[:, 0] + pcm[:, 1]
    assert np.abs(s).max() <= 32767
    np.savez_compressed(OUT, lr_sum_int16=s.astype(np.int16), sample_rate=np.int32(rate))
    mono = s.astype(np.float32) / np.float32(65536.0)
    assert np.array_equal(mono, audio[:N].mean(axis=1, dtype=np.float64).astype(np.float32))
    print(f"wrote {OUT} ({os.path.getsize(OUT)} bytes): peak {np.abs(mono).max():.4f}, rms {mono.std():.5f}")


if __name__ == "__main__":
    main()
