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
