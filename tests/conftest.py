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
    config.addinivalue_line("markers", "ref: needs oracle/_ref/libvxref.so (reference object code; built where /root/reference exists)")


@pytest.fixture(scope="session")
def vrt():
    """The package (its directory name has a hyphen, hence importlib)."""
    built = os.path.join(ROOT, "vortex-raytracing_amd", "lib", "libvortex-hip.so")
    if not os.path.exists(built):
        import __graft_entry__ as g
        g.build()
    return importlib.import_module("vortex-raytracing_amd")


@pytest.fixture(scope="session")
def po():
    from oracle import pyoracle
    pyoracle.orc()
    return pyoracle


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        path = os.path.join(GOLDEN, name + ".npz")
        if not os.path.exists(path):
            pytest.skip("fixture %s missing" % name)
        with np.load(path) as z:
            return {k: z[k] for k in z.files}
    return load


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("a GPU test was selected but no HIP device is visible (there is no CPU fallback)")
    return "cuda:0"
