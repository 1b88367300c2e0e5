"""`from model import UNet` (reference code/test.py:8, code/train.py) resolved to the MI355X implementation.

Put this directory first on PYTHONPATH to run the reference scripts without editing their imports."""
from audiodenoiser_amd.model import DoubleConvLayer, DownSampleLayer, UNet, UpSampleLayer  # noqa: F401
