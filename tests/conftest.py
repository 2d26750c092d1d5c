import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(autouse=True)
def _release_device_temporaries():
    yield
    from tests import gpu_util
    gpu_util._KEEP.clear()
    # the matmul operand dtype is context state (tn_set_matmul_dtype): a DTYPE='float16' net leaves it
    # set; per-op tests that call the C-ABI directly expect the reference's float32
    from theanet_amd import device
    if device._context is not None:
        device._context.set_matmul_dtype("float32")
