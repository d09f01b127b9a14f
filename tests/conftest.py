import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _built():
    import glob
    return (os.path.exists(os.path.join(ROOT, "distributions_amd",
                                        "libdistributions_hip.so"))
            and glob.glob(os.path.join(ROOT, "distributions_amd", "_core*.so"))
            and os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")))


def pytest_configure(config):
    config.addinivalue_line(
        "markers", "gpu: needs a real MI355X (run by the driver on a GPU box)")
    if not _built():   # a checkout without the built libraries: build once
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def oracle_built():
    import oracle_lib
    oracle_lib.build() if not os.path.exists(
        os.path.join(ROOT, "oracle", "liboracle.so")) else None
    return oracle_lib
