import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _experiment_gate(monkeypatch):
    """The engine reads its RCN_* test / experiment switches once, at creation, and only with this gate set
    (engine.hip: read_knobs): a stray variable in a user's environment cannot change the product's kernel path."""
    monkeypatch.setenv("RCN_EXPERIMENT", "1")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def hip_lib():
    """The product library; built on demand (hipcc cross-compiles without a GPU)."""
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "racon_amd", "csrc"), "all"])
    from racon_amd import engine
    return engine.load_library()
