import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)

    return load


@pytest.fixture(autouse=True)
def _seed_global_generators():
    """Every test starts from the same state of torch's global generators: module default inits (`SimpleUnet(...)` draws from them, as the
    reference's nets do) and unseeded `torch.randn` calls no longer depend on which tests ran before (round 4: a learning test's outcome did)."""
    import torch
    torch.manual_seed(20260)
    yield
