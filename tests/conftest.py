import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def built_lib():
    """libeioku_hip.so, built on demand (hipcc cross-compiles without a GPU)."""
    from eioku_amd import _lib

    if not _lib.LIB_PATH.exists():
        import __graft_entry__

        __graft_entry__.build()
    return _lib.load()


@pytest.fixture(scope="session")
def gpu(built_lib):
    """Initialised HIP device + torch device for @pytest.mark.gpu tests."""
    import torch

    if not torch.cuda.is_available():
        pytest.fail("test marked gpu but no GPU is visible: the HIP path has no CPU fallback")
    from eioku_amd import _lib

    _lib.init(0)
    return torch.device("cuda:0")
