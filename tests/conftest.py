import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    need = [os.path.join(ROOT, "fujiyama-renderer_amd", "lib", "libfjgpu.so"),
            os.path.join(ROOT, "fujiyama-renderer_amd", "lib", "libfjscene.so"),
            os.path.join(ROOT, "oracle", "liboracle.so")]
    if all(os.path.exists(p) for p in need):
        return
    import __graft_entry__
    __graft_entry__.build()


@pytest.fixture(scope="session", autouse=True)
def built_libraries():
    _ensure_built()


@pytest.fixture(scope="session")
def asset_dir():
    from fujiyama_renderer_amd import workloads
    d = workloads.default_asset_dir()
    os.makedirs(d, exist_ok=True)
    return d


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
