import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_ckpt():
    """Reference-artifact fixture (tests/golden/make_fixtures.py): shipped epoch-95 weights, two
    dataset views, derived near/far/fov, recorded PSNRs."""
    return np.load(os.path.join(ROOT, "tests", "golden", "alexander50_epoch095.npz"))


@pytest.fixture(scope="session")
def golden_vec():
    """Oracle-generated input/output vectors (tests/golden/make_golden_vectors.py)."""
    return np.load(os.path.join(ROOT, "tests", "golden", "golden_render.npz"))


@pytest.fixture(scope="session")
def oracle():
    from oracle import nerf_oracle
    return nerf_oracle
