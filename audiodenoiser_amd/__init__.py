"""MI355X-native (gfx950) implementation of the AudioDenoiser hot path: STFT magnitude + U-Net forward.

Importing the package does not touch the GPU; the HIP library is loaded (and built if missing) on first use
and every entry point raises if it is unavailable — there is no CPU or eager-PyTorch fallback.
"""
__version__ = "0.1.0"

__all__ = ["UNet", "SpectrogramDataset", "audio_to_magnitude_spectrogram", "audio_to_spectrogram",
           "stft_magnitude", "per_clip_l1"]


def __getattr__(name):
    if name == "UNet":
        from .model import UNet
        return UNet
    if name == "SpectrogramDataset":
        from .data_loader import SpectrogramDataset
        return SpectrogramDataset
    if name in ("audio_to_magnitude_spectrogram", "audio_to_spectrogram", "stft_magnitude"):
        from . import stft
        return getattr(stft, name)
    if name == "per_clip_l1":
        from .loss import per_clip_l1
        return per_clip_l1
    raise AttributeError(name)
