import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """Build the oracle (and the product library if missing) once per session."""
    import oracle_lib
    oracle_lib.build()
    import gpismap_amd
    if not os.path.exists(gpismap_amd.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    # torch ships its own copy of the HIP runtime; a test that hands device tensors to the library needs torch's
    # runtime up as well.  Bring it up first (as bench.py does): initialising it late, after many library-side
    # streams exist in the process, was seen to fail with "No HIP GPUs are available".
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
    yield
