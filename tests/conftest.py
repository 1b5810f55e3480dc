import importlib
import os
import sys

import pytest

# torch ships its own HIP runtime; it must be the first one loaded in a process that also loads
# libdlco.so, or torch later finds "No HIP GPUs" (two runtimes with one SONAME).  The product
# itself never needs torch; only the multi-rank tests and bench.py --gpus N>1 do.
import torch  # noqa: F401,E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dlco():
    """The product binding (ctypes over libdlco.so)."""
    return importlib.import_module("opencv-dlco_amd")


@pytest.fixture(scope="session")
def ref():
    """The CPU oracle (test infrastructure)."""
    from oracle import ref as r
    r.lib()
    # One thread: the oracle's problems in this suite are small (F <= 608) and its OpenMP / OpenBLAS teams cost more than
    # they give there - with every core of the box, a 64-wide step took 22 ms instead of 1.3 ms and a 544-wide one 96 ms
    # instead of 57 ms (the free-run tests were 580 of the GPU suite's 860 s).  The full-width tests (ssyevr at n = 8192)
    # raise it themselves and put it back.
    r.set_threads(1)
    return r
