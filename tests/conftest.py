import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line(
        "markers", "gpu: needs a real MI355X (run by the driver on a GPU box)")


@pytest.fixture(scope="session")
def oracle_built():
    import oracle_lib
    oracle_lib.build() if not os.path.exists(
        os.path.join(ROOT, "oracle", "liboracle.so")) else None
    return oracle_lib
