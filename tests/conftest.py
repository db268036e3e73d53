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
        return False


# multi-process arrangement tests go LAST: with `pytest -x` (how the driver runs the suite) one of them failing must not
# hide the single-kernel parity tests
RUN_LAST = ("test_configs_gpu.py", "test_multirank_gpu.py")


def pytest_collection_modifyitems(config, items):
    items.sort(key=lambda item: RUN_LAST.index(os.path.basename(str(item.fspath))) + 1
               if os.path.basename(str(item.fspath)) in RUN_LAST else 0)
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _conv_dtype_is_bf16_unless_a_test_says_otherwise():
    """functional.set_conv_dtype is process-wide (like chainer.global_config.dtype): a test that switches the 3x3 convolutions
    to MXFP8 must not leak that into the tests behind it."""
    yield
    fn = sys.modules.get("rgbd_gan_amd.functional")
    if fn is not None and fn.conv_dtype() != "bf16":
        fn.set_conv_dtype("bf16")
