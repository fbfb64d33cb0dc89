import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_npz(name):
    with np.load(os.path.join(GOLDEN, name)) as f:
        return {k: f[k] for k in f.files}


@pytest.fixture(scope="session")
def scene_states():
    """{'a': reference-init state dict, 'b': deterministic perturbation of it} (numpy, reference key names)."""
    from nrhints_amd.synthetic import perturb_state
    a = load_npz("scene_a_state.npz")
    return {"a": a, "b": perturb_state(a)}
