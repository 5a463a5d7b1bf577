import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The library's and the command line's test settings (TGSF_POOL_CAP, TGSF_BATCH_BYTES, TGSF_CLEAN_TABLES ...: they force rare
# paths and small sizes) are read only under this switch -- a user's environment never changes the path a run takes.
os.environ["TGSF_DEBUG_KNOBS"] = "1"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_collection_finish(session):
    """A GPU session that also uses torch (tests/test_gpu_fullsize.py builds its batches with it): torch ships its
    own HIP runtime and must bring the device up BEFORE libtgsf.so binds the system one, or its later
    initialisation finds no device.  Nothing is touched in a session without selected GPU tests."""
    if any(item.get_closest_marker("gpu") for item in session.items):
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:
            pass
