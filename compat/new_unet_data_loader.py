"""`from new_unet_data_loader import WavToSpecDataset` (reference code/train.py:16; the module is absent upstream)."""
from audiodenoiser_amd.data_loader import WavToSpecDataset  # noqa: F401
