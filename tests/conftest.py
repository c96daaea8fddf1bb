import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/ — test infrastructure only)."""
    from oracle import cvref

    cvref.build()
    cvref.lib()
    return cvref


@pytest.fixture(scope="session")
def oracle_fm():
    """The numpy restatement of the model-fitting half of fundamentalmatrix.rs (oracle/ - test infrastructure only)."""
    from oracle import cvref_fm

    return cvref_fm


@pytest.fixture(scope="session")
def gpu_device():
    """A cvhip device on GPU 0; the HIP extension must be built and a GPU present."""
    from cybervision_amd import correlation

    dev = correlation.create_gpu_context()
    yield dev
    dev.close()
