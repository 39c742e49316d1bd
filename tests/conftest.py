import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


_NP2T = {0: torch.float32, 1: torch.float16, 2: torch.bfloat16}


def from_bits(a: np.ndarray, dtype_code: int) -> torch.Tensor:
    """Inverse of tests/golden/make_golden.py::bits."""
    dt = _NP2T[int(dtype_code)]
    if dt == torch.float32:
        return torch.from_numpy(np.ascontiguousarray(a))
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int16)).view(dt)


@pytest.fixture(scope="session")
def oracle():
    from oracle import pe_oracle
    pe_oracle.build()
    return pe_oracle
