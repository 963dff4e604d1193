import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionfinish(session, exitstatus):
    """End of a GPU session: give everything back explicitly, in order, before the interpreter starts finalising."""
    mod = sys.modules.get("romtime_amd")
    if mod is not None:
        try:
            import torch

            if torch.cuda.is_available():
                mod.shutdown()
        except Exception as exc:   # never turn a finished session into a failure
            print(f"romtime_amd.shutdown() at session end: {exc!r}", file=sys.stderr)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden_orth():
    return load_golden("orth.npz")


@pytest.fixture(scope="session")
def golden_deim():
    return load_golden("deim.npz")


@pytest.fixture(scope="session")
def golden_rom():
    return load_golden("rom.npz")


@pytest.fixture(scope="session")
def golden_heat():
    return load_golden("heat.npz")


@pytest.fixture(scope="session")
def golden_sampler():
    return load_golden("sampler.npz")


@pytest.fixture
def cpu_ops(monkeypatch):
    """Host-logic tests: swap the device operators for the oracle's arithmetic (tests/cpu_stub.py)."""
    from tests import cpu_stub

    cpu_stub.install(monkeypatch)
    return cpu_stub
