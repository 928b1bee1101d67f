import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return os.path.exists("/dev/kfd")


HAS_GPU = _has_gpu()


def pytest_collection_modifyitems(config, items):
    if HAS_GPU:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def modsx():
    import mods_amd
    mods_amd.lib()
    return mods_amd


@pytest.fixture(scope="session")
def ctx(modsx):
    c = modsx.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def small_pair():
    from mods_amd import synthetic
    return synthetic.make_pair(rows=240, cols=320, nblobs=420, seed=777)


@pytest.fixture(scope="session")
def cat_pair():
    import numpy as np
    z = np.load(os.path.join(ROOT, "tests", "golden", "cat_pair_bgr_u8.npz"))
    return z["cat"], z["cat2"], z["H_gt"]
