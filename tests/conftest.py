"""pytest configuration: markers, shared fixtures (synthetic model files, CPU oracle)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


@pytest.fixture(scope="session")
def toy_model():
    from tools.make_synth_model import ensure_model
    return ensure_model("toy", 0)


@pytest.fixture(scope="session")
def mini_model():
    from tools.make_synth_model import ensure_model
    return ensure_model("mini", 0)


@pytest.fixture(scope="session")
def small_model():
    from tools.make_synth_model import ensure_model
    return ensure_model("small", 0)


@pytest.fixture(scope="session")
def toy_oracle(toy_model):
    from oracle.pyoracle import Oracle
    o = Oracle(toy_model, n_threads=4)
    yield o
    o.close()


@pytest.fixture(scope="session")
def mini_oracle(mini_model):
    from oracle.pyoracle import Oracle
    o = Oracle(mini_model, n_threads=4)
    yield o
    o.close()
