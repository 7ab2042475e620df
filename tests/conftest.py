import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "rate: asserts a throughput floor (collected last: a slow box must not hide parity tests behind -x)")


def pytest_collection_modifyitems(config, items):
    # the rate verdict runs behind every parity test
    items.sort(key=lambda it: 1 if it.get_closest_marker("rate") else 0)


class RateFloors:
    """Throughput floors are timing, not parity: a test notes a miss here (and warns), and tests/test_zz_rates.py -- collected
    last -- fails on the notes, so that under `pytest -x` a busy box cannot turn every later parity test into "not run"."""

    def __init__(self):
        self.missed = []

    def check(self, ok, message):
        if not ok:
            import warnings
            self.missed.append(message)
            warnings.warn("rate floor missed: " + message)


_RATE_FLOORS = RateFloors()


@pytest.fixture(scope="session")
def rate_floors():
    return _RATE_FLOORS


@pytest.fixture(scope="session")
def oracle():
    import oracle_binding
    return oracle_binding.Oracle()


@pytest.fixture(scope="session")
def engine():
    from zlibstream_amd import Engine
    return Engine(0)
