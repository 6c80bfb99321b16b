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


@pytest.fixture(scope="session")
def golden():
    def _load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return _load


def pool_input(S: int) -> np.ndarray:
    """Same recipe as tests/golden/make_golden.py::pool_input (kept in sync by
    test_oracle_golden.py::test_pool_recipe_matches_golden)."""
    rng = np.random.default_rng(1300 + S)
    A = rng.random((2, 1, S, S), dtype=np.float32)
    A[0, 0, S // 3: S // 3 + S // 8, S // 2: S // 2 + S // 8] += np.float32(5.0)
    return A
