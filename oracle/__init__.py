"""CPU oracle for the MI355X hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package.  ``audiodenoiser_amd`` never does: the product path has no CPU fallback and raises when its HIP
library is missing.

Three restatements of the reference arithmetic live here:

* ``adn_oracle.c`` (built by :func:`build` into ``oracle/_build/libadn_oracle.so``): plain-C loops for
  the U-Net forward (reference ``code/model.py:7-94``), the STFT magnitude (``librosa.stft`` +
  ``magphase`` semantics of ``code/create_train_dataset.py:162-174`` / ``code/create_test_dataset.py:35-41``)
  and the loader's fp16 quantise + pad/crop (``code/data_loader.py:41-42,54-72``).
* ``unet_torch.py``: the same forward written with ``torch.nn.functional`` calls, which dispatch to the
  same ATen/oneDNN CPU kernels the reference's ``nn.Module`` would — used where the C loops are too slow
  (full-size checks, the timed CPU baseline).
* ``stft_numpy.py``: numpy restatement of ``librosa.stft``/``magphase``.

Pinning status: U-Net — pinned by ``tests/golden/unet_*.npz`` generated from the reference's own
``model.py`` (``tools/make_golden.py``).  STFT — **parity unpinned** (librosa absent; reference holds no
STFT fixtures); cross-checked against ``numpy.fft.rfft``, ``torch.stft`` and analytic known answers.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "adn_oracle.c")
_LIB = os.path.join(_HERE, "_build", "libadn_oracle.so")
_lib = None

TAP_NAMES = ("down1", "down2", "down3", "down4", "bottleneck", "up1", "up2", "up3", "up4", "out")


def build(force: bool = False) -> str:
    """Compile ``adn_oracle.c`` with gcc (OpenMP) into ``oracle/_build``; no-op when up to date."""
    os.makedirs(os.path.dirname(_LIB), exist_ok=True)
    if (not force and os.path.exists(_LIB)
            and (not os.path.exists(_SRC) or os.path.getmtime(_LIB) >= os.path.getmtime(_SRC))):
        return _LIB
    cmd = ["gcc", "-O3", "-fopenmp", "-fPIC", "-shared", "-std=gnu11", "-o", _LIB, _SRC, "-lm"]
    subprocess.run(cmd, check=True)
    return _LIB


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB)
        fp = ctypes.POINTER(ctypes.c_float)
        L.adno_unet_forward.restype = ctypes.c_int
        L.adno_unet_forward.argtypes = [ctypes.POINTER(fp), fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.POINTER(fp), ctypes.c_int]
        L.adno_free.argtypes = [ctypes.c_void_p]
        L.adno_stft_n_frames.restype = ctypes.c_long
        L.adno_stft_n_frames.argtypes = [ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.adno_stft_mag.restype = ctypes.c_int
        L.adno_stft_mag.argtypes = [fp, ctypes.c_int, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, fp]
        L.adno_quantize_pad.argtypes = [fp, ctypes.c_int, ctypes.c_int, fp, ctypes.c_int, ctypes.c_int]
        L.adno_num_threads.restype = ctypes.c_int
        L.adno_set_num_threads.argtypes = [ctypes.c_int]
        _lib = L
    return _lib


def _fp(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def float_tensor_table(sd) -> list:
    """The 118 fp32 arrays of a state dict in schema order (num_batches_tracked dropped)."""
    out = []
    for k, v in sd.items():
        if k.endswith("num_batches_tracked"):
            continue
        out.append(np.ascontiguousarray(np.asarray(v, dtype=np.float32)))
    assert len(out) == 118, len(out)
    return out


def tap_shapes(n: int, f: int, t: int):
    ch = (64, 128, 256, 512, 1024)
    hs, ws = [f], [t]
    for _ in range(4):
        hs.append(hs[-1] // 2)
        ws.append(ws[-1] // 2)
    shapes = [(n, ch[l], hs[l], ws[l]) for l in range(4)]
    shapes.append((n, 1024, hs[4], ws[4]))
    shapes += [(n, ch[l], hs[l], ws[l]) for l in (3, 2, 1, 0)]
    shapes.append((n, 1, f, t))
    return shapes


def unet_forward(sd, x: np.ndarray, acc64: bool = False, want_taps: bool = False):
    """C-loop forward. ``x`` (N,1,F,T) fp32 -> (N,1,F,T) fp32 [, dict of the ten block outputs]."""
    L = lib()
    x = np.ascontiguousarray(x, dtype=np.float32)
    n, c, f, t = x.shape
    assert c == 1
    tens = float_tensor_table(sd)
    fp = ctypes.POINTER(ctypes.c_float)
    table = (fp * len(tens))(*[_fp(a) for a in tens])
    y = np.empty((n, 1, f, t), dtype=np.float32)
    taps = (fp * 10)() if want_taps else None
    rc = L.adno_unet_forward(table, _fp(x), _fp(y), n, f, t, taps, 1 if acc64 else 0)
    if rc != 0:
        raise ValueError(f"adno_unet_forward failed rc={rc} (F,T must be >= 16)")
    if not want_taps:
        return y
    out = {}
    for name, shp, p in zip(TAP_NAMES, tap_shapes(n, f, t), taps):
        cnt = int(np.prod(shp))
        out[name] = np.ctypeslib.as_array(p, shape=(cnt,)).copy().reshape(shp)
        L.adno_free(p)
    return y, out


def stft_n_frames(length: int, n_fft: int, hop: int, center: bool) -> int:
    return int(lib().adno_stft_n_frames(length, n_fft, hop, 1 if center else 0))


def stft_mag(audio: np.ndarray, n_fft: int, hop: int, center: bool) -> np.ndarray:
    """C-loop STFT magnitude. ``audio`` (L,) or (n_clips, L) fp32 -> (..., n_fft/2+1, n_frames) fp32."""
    a = np.ascontiguousarray(audio, dtype=np.float32)
    single = a.ndim == 1
    if single:
        a = a[None]
    nc, length = a.shape
    nfr = stft_n_frames(length, n_fft, hop, center)
    if nfr <= 0:
        raise ValueError("audio shorter than n_fft")
    out = np.empty((nc, n_fft // 2 + 1, nfr), dtype=np.float32)
    rc = lib().adno_stft_mag(_fp(a), nc, length, n_fft, hop, 1 if center else 0, _fp(out))
    if rc != 0:
        raise ValueError("adno_stft_mag failed")
    return out[0] if single else out


def quantize_pad(spec: np.ndarray, target_size) -> np.ndarray:
    """fp32(fp16(spec)) cropped / zero-padded bottom-right to ``target_size`` (data_loader.py:41-72)."""
    s = np.ascontiguousarray(spec, dtype=np.float32)
    h, w = s.shape
    H, W = target_size
    out = np.empty((H, W), dtype=np.float32)
    lib().adno_quantize_pad(_fp(s), h, w, _fp(out), H, W)
    return out


def num_threads() -> int:
    return int(lib().adno_num_threads())


def set_num_threads(n: int) -> None:
    """Cap the OpenMP team of the C restatement (the timed CPU baseline uses the process's CPU share)."""
    lib().adno_set_num_threads(int(n))
