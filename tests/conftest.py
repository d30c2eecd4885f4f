import os
import sys

import pytest

try:  # tests that hold tensors on the GPU (parity slice): torch brings its own copy of the HIP runtime, and the copy that is loaded
    import torch  # noqa: F401  -- first in a process is the one that sees the devices; load it before libtracy_hip.so pulls in /opt/rocm's
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_libraries():
    """build whatever is missing or stale (no-ops otherwise): the gfx950 library cross-compiles without a GPU"""
    from tracy_amd import build as b
    b.build()
    b.build_host()
    b.build_cli()
    b.build_msa()
    import pyoracle
    pyoracle.build()
