"""Minimal RIFF/WAVE reader and writer (numpy only) for the on-the-fly wav -> spectrogram dataset.

The reference reads audio with ``librosa.load(path, sr=SAMPLE_RATE)`` (``create_train_dataset.py:214``,
``create_test_dataset.py:144``): float32 in [-1, 1), channels averaged to mono.  librosa and soundfile are not
in this image, so the container format is parsed here: PCM 8/16/24/32-bit, IEEE float 32/64, plain and
WAVE_FORMAT_EXTENSIBLE headers.  Integer PCM is scaled by 2^-(bits-1) (8-bit: (x-128)/128), which is what
libsndfile does.  Resampling (librosa's ``sr=`` argument uses soxr) is NOT provided: files must already be at
the rate the caller expects, and ``read_wav`` returns the file's rate so callers can check.
"""
from __future__ import annotations

import struct

import numpy as np

__all__ = ["read_wav", "write_wav"]


def read_wav(path, mono: bool = True):
    """-> (audio float32 (L,) if mono else (L, channels), sample_rate)."""
    with open(path, "rb") as fh:
        blob = fh.read()
    if len(blob) < 12 or blob[:4] != b"RIFF" or blob[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(blob):
        tag, size = blob[pos:pos + 4], struct.unpack_from("<I", blob, pos + 4)[0]
        body = blob[pos + 8:pos + 8 + size]
        if tag == b"fmt ":
            fmt = body
        elif tag == b"data":
            data = body
            if fmt is not None:
                break
        pos += 8 + size + (size & 1)               # chunks are word aligned
    if fmt is None or data is None or len(fmt) < 16:
        raise ValueError(f"{path}: missing fmt or data chunk")
    code, channels, rate, _, block_align, bits = struct.unpack_from("<HHIIHH", fmt, 0)
    if code == 0xFFFE and len(fmt) >= 26:          # WAVE_FORMAT_EXTENSIBLE: real code = first 2 bytes of the GUID
        code = struct.unpack_from("<H", fmt, 24)[0]
    if channels < 1:
        raise ValueError(f"{path}: zero channels")
    bps = bits // 8
    n = len(data) // (bps * channels) * channels
    raw = np.frombuffer(data, dtype=np.uint8, count=n * bps)
    if code == 1:
        if bits == 8:
            x = (raw.astype(np.float32) - 128.0) / 128.0
        elif bits == 16:
            x = raw.view("<i2").astype(np.float32) / 32768.0
        elif bits == 24:
            b = raw.reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            v = np.where(v >= 1 << 23, v - (1 << 24), v)
            x = v.astype(np.float32) / float(1 << 23)
        elif bits == 32:
            x = (raw.view("<i4").astype(np.float64) / float(1 << 31)).astype(np.float32)
        else:
            raise ValueError(f"{path}: unsupported PCM width {bits}")
    elif code == 3:
        if bits == 32:
            x = raw.view("<f4").astype(np.float32)
        elif bits == 64:
            x = raw.view("<f8").astype(np.float32)
        else:
            raise ValueError(f"{path}: unsupported float width {bits}")
    else:
        raise ValueError(f"{path}: unsupported WAVE format code {code}")
    x = x.reshape(-1, channels)
    if mono:
        x = x[:, 0] if channels == 1 else x.mean(axis=1, dtype=np.float32)
    return np.ascontiguousarray(x, dtype=np.float32), int(rate)


def write_wav(path, audio, sample_rate: int, subtype: str = "PCM_16"):
    """Write mono/multi-channel float audio as PCM_16 or FLOAT (test fixtures and tools)."""
    a = np.asarray(audio, dtype=np.float32)
    if a.ndim == 1:
        a = a[:, None]
    channels = a.shape[1]
    if subtype == "PCM_16":
        code, bits = 1, 16
        payload = np.clip(np.rint(a * 32768.0), -32768, 32767).astype("<i2").tobytes()
    elif subtype == "FLOAT":
        code, bits = 3, 32
        payload = a.astype("<f4").tobytes()
    else:
        raise ValueError("subtype must be PCM_16 or FLOAT")
    block = channels * bits // 8
    fmt = struct.pack("<HHIIHH", code, channels, int(sample_rate), int(sample_rate) * block, block, bits)
    pad = b"\x00" if len(payload) & 1 else b""
    with open(path, "wb") as fh:
        fh.write(b"RIFF" + struct.pack("<I", 4 + 8 + len(fmt) + 8 + len(payload) + len(pad)) + b"WAVE")
        fh.write(b"fmt " + struct.pack("<I", len(fmt)) + fmt)
        fh.write(b"data" + struct.pack("<I", len(payload)) + payload + pad)
