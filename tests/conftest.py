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


def grad_bound(ref32, ref64, factor=3.0):
    """Tolerance for one gradient tensor of a training-step fixture, DERIVED from the fixture: three times
    the reference's own float32-vs-float64 distance on that tensor (its sampler places samples at fp32 noise, which the
    backward amplifies to 1-3 % on the SDF layers of scene b and to 5e-7 on the last reflectance layers), floored at 1e-4 of
    the tensor's scale.  ``factor``: 3 everywhere except the 32-ray one-hint fixtures, whose single-draw noise estimate per tensor is
    coarser (4).  Returns (absolute bound, scale); the comparison is made against the float64 gradient."""
    ref32, ref64 = np.asarray(ref32, dtype=np.float64), np.asarray(ref64, dtype=np.float64)
    scale = max(float(np.abs(ref64).max()), 1e-12)
    return max(factor * float(np.abs(ref32 - ref64).max()), 1e-4 * scale), scale
