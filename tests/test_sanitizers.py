"""Sanitizer builds of the C++ host code (race detection / memory safety): pna_host.cpp + pna_archive.cpp compiled with gcc against
a CPU stand-in for the HIP runtime (tests/san/hip/hip_runtime.h) and with the kernels stubbed at the launch layer
(tests/san/device_stub.cpp), then tests/san/san_driver.cpp is run under AddressSanitizer + UBSan and under ThreadSanitizer: container
writer, sanitize, split / join, password hashes, the batch call, the CompressionWriter facade from 24 threads (group commit, slab pool),
the bounded host pipeline with its stager thread, the streaming entry writer, append.  No GPU involved."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

SAN = os.path.join(ROOT, "tests", "san")


@pytest.mark.parametrize("target", ["asan", "tsan"])
def test_host_code_under_sanitizers(target):
    if shutil.which("g++") is None:
        pytest.fail("g++ is needed for the sanitizer build of the host code")
    r = subprocess.run(["make", "-s", "-C", SAN, target], capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "all checks passed" in r.stdout and "ERROR: " not in tail and "WARNING: ThreadSanitizer" not in tail, tail
