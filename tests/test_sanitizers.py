"""an ASan/UBSan target for the host code.  `make -C tests/shim asan` builds tests/shim/asan_main.cpp
(pair indexing, I/O layer, export format - the product's host-side headers - and the C oracle) with
-fsanitize=address,undefined and runs it: any report fails the make.  GPU AddressSanitizer is not available on this pool."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_code_under_asan_and_ubsan():
    if shutil.which("g++") is None or shutil.which("make") is None:
        pytest.skip("no host toolchain")
    p = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "shim"), "asan"], capture_output=True, text=True, timeout=900)
    out = p.stdout + p.stderr
    if p.returncode != 0 and ("cannot find -lasan" in out or "libasan" in out and "No such file" in out):
        pytest.skip("libasan is not installed on this machine")
    assert p.returncode == 0 and "all host checks passed" in out, out[-4000:]
    assert "runtime error" not in out and "AddressSanitizer" not in out
