import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def O():
    import oracle

    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def wtp():
    import wtp_amd

    return wtp_amd


@pytest.fixture(scope="session")
def ctx(wtp):
    """One libwtp context for the whole GPU session; creating it fails loudly without the .so
    or without a gfx950 device (there is no CPU path to fall back to)."""
    c = wtp.Context(0)
    yield c
    c.close()
