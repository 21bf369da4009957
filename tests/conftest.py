import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", "reference_numpy_half.npz")
    return np.load(path, allow_pickle=False)


@pytest.fixture(autouse=True)
def _fixed_action_noise_key():
    """Tests that compare two runs of a rollout need the same action noise in both: the sampling key (drawn from
    the OS when nobody sets it) is pinned before every test."""
    try:
        from baselines_amd.ppo2 import policies
    except Exception:       # library not built: the tests that need it fail on their own import
        yield
        return
    policies.set_sample_seed(0)
    policies.set_sample_rank(0)
    yield
