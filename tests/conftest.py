import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes tens of seconds on the CPU")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture
def setenv(monkeypatch):
    """setenv(name, value) for the library's MCAMD_* switches: they are cached at first use (never read per launch), so
    a test that changes one has the library re-read them, and again when the change is undone."""
    from modelcompression_amd import _lib

    def _set(name, value):
        monkeypatch.setenv(name, value)
        _lib.reload_config()
    yield _set
    monkeypatch.undo()
    _lib.reload_config()
