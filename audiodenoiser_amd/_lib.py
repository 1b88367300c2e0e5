"""ctypes binding of libadn.so (include/adn.h).  There is no fallback: if the HIP library cannot be loaded
or built, every entry point raises."""
from __future__ import annotations

import ctypes
import os
import threading

from . import build as _build

_lock = threading.Lock()
_lib = None

c_float_p = ctypes.POINTER(ctypes.c_float)


class AdnError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load (building first if the in-tree library is missing or stale) libadn.so."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        variant = os.environ.get("ADN_LIBADN_PATH")       # variant sweeps of tools/ (build.build_variant); unset in production
        try:
            path = variant if variant else _build.build()
        except Exception as exc:  # noqa: BLE001 - re-raised with context
            raise AdnError(f"libadn.so (MI355X HIP kernels) is missing and could not be built: {exc}") from exc
        # PyTorch-ROCm bundles its own HIP runtime (torch/lib/libamdhip64.so, soname libamdhip64.so.7).  It must be
        # in the process BEFORE libadn.so is mapped so that libadn's NEEDED libamdhip64.so.7 binds to that same
        # runtime instance (streams and allocations are shared with torch); loading libadn first would pull in
        # /opt/rocm's copy and leave two HIP runtimes in one process.
        import torch  # noqa: F401
        try:
            L = ctypes.CDLL(path)
        except OSError as exc:
            raise AdnError(f"cannot load {path}: {exc}") from exc
        vp, sz, ci, cl = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_long
        L.adn_version.restype = ci
        L.adn_last_error.restype = ctypes.c_char_p
        L.adn_device_count.argtypes = [ctypes.POINTER(ci)]
        L.adn_prepare.argtypes = [ci, ci]
        L.adn_unet_create.argtypes = [ctypes.POINTER(vp), ci, ctypes.POINTER(c_float_p), ci]
        L.adn_unet_create_ex.argtypes = [ctypes.POINTER(vp), ci, ctypes.POINTER(c_float_p), ci, ci]
        L.adn_unet_create_general.argtypes = [ctypes.POINTER(vp), ci, ctypes.POINTER(c_float_p), ci, ci, ci, ci]
        L.adn_unet_channels.argtypes = [vp, ctypes.POINTER(ci), ctypes.POINTER(ci)]
        L.adn_unet_set_batch_invariant.argtypes = [vp, ci]
        L.adn_unet_destroy.argtypes = [vp]
        L.adn_unet_workspace_bytes.argtypes = [vp, ci, ci, ci, ctypes.POINTER(sz)]
        L.adn_unet_forward.argtypes = [vp, vp, vp, ci, ci, ci, vp, sz, vp]
        L.adn_unet_forward_taps.argtypes = [vp, vp, vp, ci, ci, ci, vp, sz, ctypes.POINTER(vp), vp]
        L.adn_unet_set_timing.argtypes = [vp, ci]
        L.adn_unet_get_timing.argtypes = [vp, ci, c_float_p]
        L.adn_stft_n_frames.argtypes = [cl, ci, ci, ci, ctypes.POINTER(cl)]
        L.adn_stft_mag.argtypes = [vp, ci, cl, ci, ci, ci, vp, vp]
        L.adn_stft_mag_fit.argtypes = [vp, ci, cl, ci, ci, ci, vp, ci, ci, vp]
        L.adn_quantize_pad.argtypes = [vp, ci, ci, ci, vp, ci, ci, vp]
        L.adn_per_clip_l1.argtypes = [vp, vp, ci, cl, vp, vp]
        L.adn_perceptual_loss_workspace_bytes.argtypes = [ci, ci, ci, ctypes.POINTER(sz)]
        L.adn_perceptual_loss.argtypes = [vp, vp, ci, ci, ci, vp, sz, vp, vp]
        L.adn_istft_length.argtypes = [ci, ci, ctypes.POINTER(cl)]
        L.adn_griffin_lim_workspace_bytes.argtypes = [ci, ci, ci, ctypes.POINTER(sz)]
        L.adn_griffin_lim.argtypes = [vp, vp, ci, ci, ci, ci, ci, ci, vp, sz, vp, vp]
        L.adn_stft_complex.argtypes = [vp, ci, cl, ci, ci, vp, vp]
        L.adn_istft_workspace_bytes.argtypes = [ci, ci, ci, ctypes.POINTER(sz)]
        L.adn_istft.argtypes = [vp, ci, ci, ci, ci, vp, sz, vp, vp]
        for name in ("adn_device_count", "adn_prepare", "adn_unet_create", "adn_unet_create_ex", "adn_unet_create_general", "adn_unet_channels",
                     "adn_unet_set_batch_invariant", "adn_unet_destroy", "adn_unet_workspace_bytes", "adn_unet_forward", "adn_unet_forward_taps", "adn_unet_set_timing", "adn_unet_get_timing",
                     "adn_stft_n_frames", "adn_stft_mag", "adn_stft_mag_fit", "adn_quantize_pad", "adn_per_clip_l1",
                     "adn_perceptual_loss_workspace_bytes", "adn_perceptual_loss", "adn_istft_length",
                     "adn_griffin_lim_workspace_bytes", "adn_griffin_lim", "adn_stft_complex",
                     "adn_istft_workspace_bytes", "adn_istft"):
            getattr(L, name).restype = ci
        _lib = L
        return L


def staging_device():
    """The ROCm device host tensors are staged on when a caller hands CPU tensors to the mirror (the reference's
    ``test.py`` keeps model and data on the CPU, ``test.py:63-66,100,112-113``): torch's current device.  There is
    no CPU implementation, so without a device this raises."""
    import torch
    if not torch.cuda.is_available():
        raise AdnError("audiodenoiser_amd: no ROCm device is visible; the hot path has no CPU implementation "
                       "(CPU tensors are staged onto the current ROCm device, computed there by libadn.so and copied back)")
    return torch.device("cuda", torch.cuda.current_device())


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().adn_last_error()
        raise AdnError(f"{what} failed (status {rc}): {msg.decode() if msg else ''}")


EXPORTED_SYMBOLS = (
    "adn_version", "adn_last_error", "adn_device_count", "adn_prepare", "adn_unet_create", "adn_unet_create_ex", "adn_unet_create_general",
    "adn_unet_channels", "adn_unet_set_batch_invariant", "adn_unet_destroy", "adn_unet_workspace_bytes", "adn_unet_forward", "adn_unet_forward_taps", "adn_unet_set_timing",
    "adn_unet_get_timing", "adn_stft_n_frames", "adn_stft_mag", "adn_stft_mag_fit",
    "adn_quantize_pad", "adn_per_clip_l1", "adn_perceptual_loss_workspace_bytes", "adn_perceptual_loss",
    "adn_istft_length", "adn_griffin_lim_workspace_bytes", "adn_griffin_lim", "adn_stft_complex",
    "adn_istft_workspace_bytes", "adn_istft",
)
