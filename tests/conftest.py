import os
import sys

import pytest

try:  # torch bundles its own HIP runtime: when both live in one process torch must be loaded first
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    """the host package (its directory name is not a Python identifier, so it is loaded by path)"""
    from __graft_entry__ import load_package
    return load_package()


@pytest.fixture(scope="session")
def orc():
    import oracle_lib
    return oracle_lib.Oracle()


@pytest.fixture(scope="session")
def ctx(pkg):
    c = pkg.Context(0)  # raises loudly when there is no gfx950 device: GPU tests must not pass silently
    yield c
    c.close()


@pytest.fixture(scope="session")
def golden():
    return os.path.join(ROOT, "tests", "golden")
