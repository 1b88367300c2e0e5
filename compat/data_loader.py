"""`from data_loader import SpectrogramDataset` (reference code/data_loader.py) resolved to the MI355X mirror."""
from audiodenoiser_amd.data_loader import SpectrogramDataset, WavToSpecDataset  # noqa: F401
