import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def headline_context(pna, flags=None, device=0):
    """A context PINNED to the headline form of the encoder -- 128 KiB blocks, whole segments per workgroup, i.e. the library's latency mode for small
    batches switched off --: what most parity tests compare with the model at its default block size.  Tests OPT IN by calling this (or taking the
    `gpu_ctx` fixture, which is such a context); a test that creates `pna.Context(0)` itself runs the library's defaults, latency mode included
    (tests/test_gpu_latency.py, the full-size cases, the seam).  (Until round 4 an autouse fixture pinned this through the environment for every test.)"""
    ctx = pna.Context(device) if flags is None else pna.Context(device, flags=flags)
    ctx.set_option("latency_max_mib", 0)
    return ctx


@pytest.fixture(scope="session")
def codec():
    from oracle import codec as c
    c.lib()  # builds oracle/liboracle.so on first use
    return c


@pytest.fixture(scope="session")
def pf():
    from oracle import pna_format
    return pna_format


@pytest.fixture(scope="session")
def pna():
    """The product package; building it needs hipcc (present here and on the GPU box)."""
    lib = os.path.join(ROOT, "portable-network-archive_amd", "libpna_gpu.so")
    if not os.path.exists(lib):
        import __graft_entry__ as g
        g.build()
    return importlib.import_module("portable-network-archive_amd")


@pytest.fixture(scope="session")
def gpu_ctx(pna):
    import torch  # noqa: F401  (shares its HIP runtime with the extension)
    ctx = headline_context(pna)
    yield ctx
    ctx.close()


@pytest.fixture
def big_ctx(pna, monkeypatch):
    """A context of its own for the full-size cases: their multi-GiB workspaces (decoder scratch, staging) are released with it instead
    of staying in the session's context, and torch's cached blocks are handed back before and after.  The full-size cases run the
    library's defaults."""
    import torch
    torch.cuda.empty_cache()
    ctx = pna.Context(0)
    yield ctx
    ctx.close()
    torch.cuda.empty_cache()


def golden(name: str) -> bytes:
    with open(os.path.join(GOLDEN, name), "rb") as f:
        return f.read()
