import os
import sys

import numpy as np
import pytest

# the captured step loop (plnlp_amd/capture.py) is opt-in: the product only flips ROCm's graph-packet-capture switch when
# PLNLP_CAPTURE=1 is in the environment at import.  The suite TESTS that loop (tests/test_hip_round3.py), so the test
# session asks for it here, before anything has touched the GPU -- a choice of this test environment, not of the library.
os.environ.setdefault("PLNLP_CAPTURE", "available")     # the runtime flag only: the default loop stays eager

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracle (and torch's CPU ops in general) must stay inside the container's CPU quota: with one thread
    # per visible core the GPU boxes (256 cores shown, 16 granted) freeze the process for most of every 100 ms
    from plnlp_amd.utils import limit_host_threads
    limit_host_threads()


def pytest_collection_modifyitems(config, items):
    """GPU tests must fail loudly (not skip) when asked for explicitly with -m gpu on a
    box without the HIP library / device; without -m they are deselected by the driver's
    `-m "not gpu"`.
    Order: tests/test_hip_multirank.py first -- it starts child processes that use the GPU, and the parent should not have
    initialised the GPU itself when it does (stable sort: everything else keeps its order)."""
    items.sort(key=lambda it: 0 if "test_hip_multirank" in it.nodeid else 1)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load
