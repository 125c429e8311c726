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


@pytest.fixture(autouse=True)
def _split_lz_stage_for_small_batches(monkeypatch):
    """The suite pins the split form of the LZ stage (k_lzm + k_lzp: the library's default for every run since the parse kernel runs one wave per
    block; the option only matters if a threshold is configured); the one-kernel form has tests of its own (test_lz_stage_forms_are_identical,
    the latency mode's units)."""
    if "PNA_LZ_SPLIT_MIN" not in os.environ:
        monkeypatch.setenv("PNA_LZ_SPLIT_MIN", "0")
    # The suite switches the LATENCY MODE off (small batches cut into small blocks and LZ units: other kernels' paths, other
    # model parameters): the headline path is what most tests pin; tests/test_gpu_latency.py covers the mode itself and the library's default.
    if "PNA_LATENCY_MAX_MIB" not in os.environ:
        monkeypatch.setenv("PNA_LATENCY_MAX_MIB", "0")
    # (pna_gpu_init reads both once: contexts are created inside the tests, after this fixture)


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
    ctx = pna.Context(0)
    # (a session fixture is set up before the function-scoped one above has touched the environment: say it here)
    if "PNA_LZ_SPLIT_MIN" not in os.environ or os.environ["PNA_LZ_SPLIT_MIN"] == "0":
        ctx.set_option("lz_split_min", 0)
    if "PNA_LATENCY_MAX_MIB" not in os.environ or os.environ["PNA_LATENCY_MAX_MIB"] == "0":
        ctx.set_option("latency_max_mib", 0)
    yield ctx
    ctx.close()


@pytest.fixture
def big_ctx(pna, monkeypatch):
    """A context of its own for the full-size cases: their multi-GiB workspaces (decoder scratch, staging) are released with it instead
    of staying in the session's context, and torch's cached blocks are handed back before and after.  The full-size cases run the
    library's defaults (see _split_lz_stage_for_small_batches)."""
    import torch
    monkeypatch.delenv("PNA_LZ_SPLIT_MIN", raising=False)
    monkeypatch.delenv("PNA_LATENCY_MAX_MIB", raising=False)
    torch.cuda.empty_cache()
    ctx = pna.Context(0)
    yield ctx
    ctx.close()
    torch.cuda.empty_cache()


def golden(name: str) -> bytes:
    with open(os.path.join(GOLDEN, name), "rb") as f:
        return f.read()
