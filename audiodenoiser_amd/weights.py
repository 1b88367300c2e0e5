"""State-dict schema of the reference U-Net and a portable synthetic-weight generator.

The reference ships no trained checkpoint (``/root/reference/code/test.py:15,59`` expects
``./saved_models/*.pth``, absent), and its 31 042 369 parameters (124 MB fp32) are too large to commit
as a fixture.  Parity fixtures, tests and ``bench.py`` therefore regenerate identical weights anywhere
from a counter-based hash PRNG keyed by ``(seed, state_dict key, flat element index)``; only numpy
integer arithmetic is involved, so the stream is bit-identical on every host.

Schema source: ``/root/reference/code/model.py:7-17`` (DoubleConvLayer = conv3x3, BN, ReLU, conv3x3, BN,
ReLU), ``:23-32`` (DownSampleLayer), ``:35-50`` (UpSampleLayer: ConvTranspose2d k2 s2 + DoubleConv on
2*C channels), ``:53-68`` (UNet: 1-64-128-256-512-[1024]-512-256-128-64-1).
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

BN_EPS = 1e-5  # nn.BatchNorm2d default, reference model.py:12,15

# (block name, Cin, Cout) of the nine DoubleConvLayer instances, in module order (model.py:56-66).
DOUBLE_CONVS = (
    ("downconv1.conv", 1, 64),
    ("downconv2.conv", 64, 128),
    ("downconv3.conv", 128, 256),
    ("downconv4.conv", 256, 512),
    ("bottleneck", 512, 1024),
    ("upconv1.conv", 1024, 512),
    ("upconv2.conv", 512, 256),
    ("upconv3.conv", 256, 128),
    ("upconv4.conv", 128, 64),
)
# (block name, Cin, Cout) of the four ConvTranspose2d(k=2, s=2) layers (model.py:38).
UP_CONVTS = (
    ("upconv1.up", 1024, 512),
    ("upconv2.up", 512, 256),
    ("upconv3.up", 256, 128),
    ("upconv4.up", 128, 64),
)


def state_dict_schema(in_channels: int = 1, num_classes: int = 1) -> "OrderedDict[str, tuple]":
    """Ordered ``key -> shape`` of ``UNet(in_channels, num_classes).state_dict()`` (136 entries).

    Order follows torch's module registration order of the reference ``model.py:53-68``:
    downconv1..4 (pool has no state, then ``conv``), bottleneck, upconv1..4 (``up`` then ``conv``), out.
    """
    sd: "OrderedDict[str, tuple]" = OrderedDict()

    def double_conv(prefix, cin, cout):
        for idx, ci in ((0, cin), (3, cout)):
            sd[f"{prefix}.double_conv.{idx}.weight"] = (cout, ci, 3, 3)
            sd[f"{prefix}.double_conv.{idx}.bias"] = (cout,)
            bn = idx + 1
            sd[f"{prefix}.double_conv.{bn}.weight"] = (cout,)
            sd[f"{prefix}.double_conv.{bn}.bias"] = (cout,)
            sd[f"{prefix}.double_conv.{bn}.running_mean"] = (cout,)
            sd[f"{prefix}.double_conv.{bn}.running_var"] = (cout,)
            sd[f"{prefix}.double_conv.{bn}.num_batches_tracked"] = ()

    ups = {name.split(".")[0]: (cin, cout) for name, cin, cout in UP_CONVTS}
    for name, cin, cout in DOUBLE_CONVS:
        top = name.split(".")[0]
        if name == "downconv1.conv":
            cin = in_channels
        if top in ups:
            ucin, ucout = ups[top]
            sd[f"{top}.up.weight"] = (ucin, ucout, 2, 2)
            sd[f"{top}.up.bias"] = (ucout,)
        double_conv(name, cin, cout)
    sd["out.weight"] = (num_classes, 64, 1, 1)
    sd["out.bias"] = (num_classes,)
    return sd


# ---------------------------------------------------------------------------------------------
# counter-based PRNG
# ---------------------------------------------------------------------------------------------
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return x


def hash_uniform(seed: int, key: str, n: int, offset: int = 0) -> np.ndarray:
    """``n`` float32 uniforms in [0, 1) for stream ``(seed, key)``, elements ``offset .. offset+n``.

    u_i = (splitmix64(splitmix64(seed ^ fnv1a64(key)) + i) >> 40) * 2^-24  — exactly representable in
    fp32, so no rounding-mode dependence.
    """
    base = _splitmix64(np.array([(seed ^ _fnv1a64(key)) & 0xFFFFFFFFFFFFFFFF], dtype=np.uint64))[0]
    with np.errstate(over="ignore"):
        ctr = np.arange(offset, offset + n, dtype=np.uint64) + base
    bits = _splitmix64(ctr) >> np.uint64(40)
    return (bits.astype(np.float32)) * np.float32(2.0 ** -24)


def make_state_dict(seed: int = 1234, in_channels: int = 1, num_classes: int = 1) -> "OrderedDict[str, np.ndarray]":
    """Synthetic, signal-preserving weights for the reference schema (numpy, fp32 / int64).

    * conv3x3 / conv1x1 weights: uniform(+-sqrt(6/fan_in)) (variance-preserving through ReLU, so every
      layer still matters at the output — torch's default init makes the output nearly constant and
      hides errors);  convT weights: uniform(+-sqrt(3/Cin)).
    * conv biases: uniform(+-0.1).
    * BatchNorm: running_mean (u-0.5)*0.2, running_var 0.75+0.5u, weight 0.75+0.5u, bias (u-0.5)*0.2
      (non-trivial statistics so a wrong BN fold cannot pass), num_batches_tracked = 100.
    """
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for key, shape in state_dict_schema(in_channels, num_classes).items():
        n = int(np.prod(shape)) if shape else 1
        if key.endswith("num_batches_tracked"):
            sd[key] = np.array(100, dtype=np.int64)
            continue
        u = hash_uniform(seed, key, n)
        leaf = key.rsplit(".", 1)[1]
        is_bn = len(shape) == 1 and (".double_conv.1." in key or ".double_conv.4." in key)
        if is_bn:
            if leaf == "running_mean" or leaf == "bias":
                v = (u - np.float32(0.5)) * np.float32(0.2)
            else:  # running_var, weight
                v = np.float32(0.75) + np.float32(0.5) * u
        elif leaf == "bias":
            v = (u - np.float32(0.5)) * np.float32(0.2)
        elif ".up." in key:  # ConvTranspose2d (Cin, Cout, 2, 2): one tap per output, fan_in = Cin
            bound = np.float32(np.sqrt(3.0 / shape[0]))
            v = (u * np.float32(2.0) - np.float32(1.0)) * bound
        else:  # Conv2d (Cout, Cin, kh, kw)
            fan_in = shape[1] * shape[2] * shape[3]
            bound = np.float32(np.sqrt(6.0 / fan_in))
            v = (u * np.float32(2.0) - np.float32(1.0)) * bound
        sd[key] = v.astype(np.float32).reshape(shape)
    return sd


WEIGHT_VARIANTS = ("benign", "trained", "heavy")


def make_state_dict_variant(kind: str, seed: int = 1234, in_channels: int = 1,
                            num_classes: int = 1) -> "OrderedDict[str, np.ndarray]":
    """Parameter distributions a real checkpoint could have and ``make_state_dict`` (kind "benign") never shows the kernels.

    * ``"trained"``: torch-default-scale convolutions (``kaiming_uniform_(a=sqrt(5))``: weights and biases uniform
      (+-1/sqrt(fan_in)); ConvTranspose2d uses torch's fan_in = Cout*4) and "trained-like" BatchNorm — ``running_var``
      log-uniform in [1e-3, 1e2], ``weight`` (gamma) uniform in [-1.5, 1.5] (negative and near-zero gammas included),
      ``bias`` and ``running_mean`` uniform in [-1, 1].  The folded per-channel scale gamma/sqrt(var+eps) spans 0 .. 47.
    * ``"heavy"``: the benign weights with heavy tails — in every convolution / transposed-convolution weight tensor
      8 hash-chosen elements are multiplied by 20 (outliers the Winograd weight transform G g G^T spreads over a whole
      6x6 block).
    Same counter-based PRNG as ``make_state_dict``: bit-identical on every host.
    """
    if kind == "benign":
        return make_state_dict(seed, in_channels, num_classes)
    if kind not in WEIGHT_VARIANTS:
        raise ValueError(f"unknown weight variant {kind!r}; expected one of {WEIGHT_VARIANTS}")
    if kind == "heavy":
        sd = make_state_dict(seed, in_channels, num_classes)
        for key, v in sd.items():
            if v.ndim == 4:
                u = hash_uniform(seed, "outlier:" + key, 8).astype(np.float64)
                idx = np.minimum((u * v.size).astype(np.int64), v.size - 1)
                flat = v.reshape(-1)
                flat[idx] = flat[idx] * np.float32(20.0)
        return sd
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    two = np.float32(2.0)
    one = np.float32(1.0)
    last_fan_in = 1
    for key, shape in state_dict_schema(in_channels, num_classes).items():
        n = int(np.prod(shape)) if shape else 1
        if key.endswith("num_batches_tracked"):
            sd[key] = np.array(100, dtype=np.int64)
            continue
        u = hash_uniform(seed, "trained:" + key, n)
        leaf = key.rsplit(".", 1)[1]
        is_bn = len(shape) == 1 and (".double_conv.1." in key or ".double_conv.4." in key)
        if is_bn:
            if leaf == "running_var":
                # log-uniform-like over 2^-10 .. 2^6 * 1.64 (9.8e-4 .. 105) from IEEE-exact operations only (no libm pow, whose last
                # bit may differ between hosts): var = 2^floor(e) * (1 + frac(e)), e = -10 + 16.64 u
                e = np.float64(-10.0) + np.float64(16.64) * u.astype(np.float64)
                k = np.floor(e)
                v = np.ldexp(1.0 + (e - k), k.astype(np.int64)).astype(np.float32)
            elif leaf == "weight":
                v = (u * two - one) * np.float32(1.5)
            else:  # bias, running_mean
                v = u * two - one
        elif len(shape) == 4:
            # torch: fan_in = size(1) * receptive field (for ConvTranspose2d's (Cin, Cout, 2, 2) that is Cout * 4)
            last_fan_in = shape[1] * shape[2] * shape[3]
            v = (u * two - one) * np.float32(1.0 / np.sqrt(last_fan_in))
        else:  # bias of the convolution just defined
            v = (u * two - one) * np.float32(1.0 / np.sqrt(last_fan_in))
        sd[key] = np.asarray(v, dtype=np.float32).reshape(shape)
    return sd


def make_input(seed: int, n: int, f: int, t: int, scale: float = 3.0) -> np.ndarray:
    """Synthetic non-negative magnitude-like input ``(n, 1, f, t)`` fp32 = uniform[0,1) * scale."""
    u = hash_uniform(seed, "input", n * f * t)
    return (u * np.float32(scale)).reshape(n, 1, f, t)


def make_audio(seed: int, n_clips: int, length: int) -> np.ndarray:
    """Synthetic audio ``(n_clips, length)`` fp32 uniform(-1, 1)."""
    u = hash_uniform(seed, "audio", n_clips * length)
    return (u * np.float32(2.0) - np.float32(1.0)).reshape(n_clips, length)
