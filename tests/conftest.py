import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The C-ABI library is built in-tree and git-ignored: a fresh checkout has the sources only.  Build it once
    # (hipcc cross-compiles gfx950 without a GPU) instead of failing every test at import.
    if not os.path.exists(os.path.join(ROOT, "live2diff_amd", "libl2d_hip.so")):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name + ".npz")))

    return load
