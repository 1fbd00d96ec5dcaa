import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as g
    return g.load_package()


@pytest.fixture(scope="session")
def standins(pkg):
    import importlib
    return importlib.import_module("pem_spgemm_amd.standins")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle is the checker; tests are one of the three places allowed to load it."""
    import __graft_entry__ as g
    return g.load_oracle()


@pytest.fixture(scope="session")
def ctx(pkg):
    c = pkg.Context(0)
    yield c
    c.close()
