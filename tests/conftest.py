import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def weights_np():
    """Synthetic state dict (seed 1234) shared by every test in the session."""
    from audiodenoiser_amd.weights import make_state_dict
    return make_state_dict(1234)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN_DIR


def load_real_audio_fixture(golden_dir, name="real_audio_17480-2-0-24.npz"):
    """The real-audio fixtures can be deleted from a tree that must not redistribute the clip (tests/golden/README.md):
    the tests that read them skip instead of failing."""
    import numpy as np
    path = os.path.join(golden_dir, name)
    if not os.path.exists(path):
        pytest.skip(f"{name} is not in this tree (see tests/golden/README.md)")
    return np.load(path)


VARIANT_KINDS = ("trained", "heavy")
VARIANT_SHAPES = ((33, 47), (257, 188), (513, 256))


def variant_input(golden_dir, f, t):
    """Input of the weight-variant goldens (tools/make_golden.py::variant_input): the real-audio network input of
    configs[0] cropped top-left to (f, t), two clips at scale 1 and 100."""
    import numpy as np
    x16 = load_real_audio_fixture(golden_dir, "config0_real_audio.npz")["x_f16"]
    x = x16[:f, :t].astype(np.float32)
    return np.stack([x, x * np.float32(100.0)])[:, None]


def load_variant_golden(golden_dir, kind, f, t):
    import numpy as np
    path = os.path.join(golden_dir, f"unet_{kind}_{f}x{t}.npz")
    if not os.path.exists(path):
        pytest.skip(f"{os.path.basename(path)} is not in this tree")
    return np.load(path)


@pytest.fixture(scope="session")
def variant_weights():
    """kind -> state dict of make_state_dict_variant(kind, 1234), built once per session (124 MB each)."""
    from audiodenoiser_amd.weights import make_state_dict_variant
    cache = {}

    def get(kind):
        if kind not in cache:
            cache[kind] = make_state_dict_variant(kind, 1234)
        return cache[kind]
    return get
