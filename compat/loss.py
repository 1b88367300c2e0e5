"""`from loss import CombinedPerceptualLoss` (reference code/loss.py) resolved to the device implementation
(evaluation only: per-clip kernels, no autograd)."""
from audiodenoiser_amd.loss import CombinedPerceptualLoss  # noqa: F401
