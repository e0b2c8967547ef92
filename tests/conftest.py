import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with `-m gpu` on the GPU box")


def pose_delta(A, B):
    """(|dt| in m, |dR| in rad) between two 4x4 rigid transforms."""
    D = np.linalg.inv(A) @ B
    R = D[:3, :3]
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) * 0.5
    ang = float(np.arctan2(np.linalg.norm(w), (np.trace(R) - 1.0) * 0.5))
    return float(np.linalg.norm(D[:3, 3])), ang


@pytest.fixture(scope="session")
def hip_module():
    """The product binding.  Fails loudly (no skip) when the HIP library or the GPU is missing."""
    from tloam_amd import registration as reg
    reg.load_library()
    return reg
