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


def _reload_pe_env():
    """The library reads its PE_* tunables once; tests that change one have it read them again."""
    lib_mod = sys.modules.get("vllm_paged_compression_amd._lib")
    if lib_mod is not None and getattr(lib_mod, "_lib", None) is not None:
        lib_mod._lib.pe_reload_env()


@pytest.fixture(autouse=True)
def _pe_env_restored_after_test():
    # set up before (so torn down after) monkeypatch: the library re-reads the restored environment
    yield
    _reload_pe_env()


@pytest.fixture
def monkeypatch(monkeypatch):
    """The built-in fixture, with setenv / delenv of PE_* variables followed by a reload of the
    library's cached tunables."""
    set_env, del_env = monkeypatch.setenv, monkeypatch.delenv

    def setenv(name, value, *a, **k):
        set_env(name, value, *a, **k)
        if name.startswith("PE_"):
            _reload_pe_env()

    def delenv(name, *a, **k):
        del_env(name, *a, **k)
        if name.startswith("PE_"):
            _reload_pe_env()

    monkeypatch.setenv, monkeypatch.delenv = setenv, delenv
    return monkeypatch
