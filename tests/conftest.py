import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the oracle (CPU) is built; libpdr_hip.so is built by __graft_entry__.build()."""
    from oracle import pdr_oracle
    pdr_oracle.build()
    lib = os.path.join(ROOT, "point_diffusion_refinement_amd", "libpdr_hip.so")
    if not os.path.exists(lib):
        import __graft_entry__
        __graft_entry__.build()
    yield


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def pytest_sessionfinish(session, exitstatus):
    """Measured floating-point parity of this session's HIP-vs-reference comparisons (tests/parity.py) ->
    gpurun_out/parity.json (merged by comparison name; committed as profiles/r3_parity.json)."""
    try:
        from tests import parity
        parity.dump(os.path.join(ROOT, "gpurun_out", "parity.json"))
    except Exception as e:                                    # bookkeeping must never fail a run
        print("parity dump skipped:", e)
