import os
import sys

import numpy as np
import pytest

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
    `-m "not gpu"`."""
    return


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load
