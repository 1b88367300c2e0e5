"""MI355X-native (gfx950) implementation of the AudioDenoiser hot path: STFT magnitude + U-Net forward.

Importing the package does not touch the GPU; the HIP library is loaded (and built if missing) on first use
and every entry point raises if it is unavailable — there is no CPU or eager-PyTorch fallback.
"""
__version__ = "0.1.0"

__all__ = ["UNet", "SpectrogramDataset", "WavToSpecDataset", "audio_to_magnitude_spectrogram",
           "audio_to_spectrogram", "stft_magnitude", "per_clip_l1", "CombinedPerceptualLoss"]


def __getattr__(name):
    if name == "UNet":
        from .model import UNet
        return UNet
    if name in ("SpectrogramDataset", "WavToSpecDataset"):
        from . import data_loader
        return getattr(data_loader, name)
    if name in ("audio_to_magnitude_spectrogram", "audio_to_spectrogram", "stft_magnitude"):
        from . import stft
        return getattr(stft, name)
    if name in ("per_clip_l1", "CombinedPerceptualLoss"):
        from . import loss
        return getattr(loss, name)
    raise AttributeError(name)
