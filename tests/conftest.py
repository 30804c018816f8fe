import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def B():
    """the product binding; libmipt.so must already be built in-tree"""
    pkg = importlib.import_module("path-tracer_amd")
    pkg.binding.lib()
    return pkg.binding


@pytest.fixture(scope="session")
def ob():
    import oracle_binding
    oracle_binding.lib()
    return oracle_binding


@pytest.fixture(scope="session")
def sf():
    import scene_fixtures
    return scene_fixtures
