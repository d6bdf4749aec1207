import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built_library():
    """The HIP library, built in-tree (cross-compiles without a GPU)."""
    from unityraytracer_amd import build
    return build.build_library()


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import pyoracle
    return pyoracle.load()


@pytest.fixture(scope="session")
def gpu_ctx(built_library):
    """One Context for the whole GPU session (one process on the card)."""
    from unityraytracer_amd import Context
    ctx = Context(0)
    yield ctx
    ctx.close()
