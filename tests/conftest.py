import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def scene_arrays_from_oracle(osc):
    """Interchange arrays of an oracle scene: (spheres9, material_id, mats6, nodes8, camera21)."""
    from oracle import camera24_to_abi

    sp, mt, nd, cam24 = osc.export()
    mid = np.arange(sp.shape[0], dtype=np.uint32)
    return sp, mid, mt, nd, camera24_to_abi(cam24)


@pytest.fixture(scope="session")
def gpu_ctx():
    import rrt_amd

    ctx = rrt_amd.Context(0)
    yield ctx
    ctx.close()
