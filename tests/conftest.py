import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "experimental: needs a GPU AND the experimental library (BP_LIB=.../libbetapose_hip_exp.so): "
                                       "kernels that were measured and superseded, not part of the product")


def pytest_collection_modifyitems(config, items):
    """Tests marked ``gpu`` (including the ones that only drive subprocesses) skip on a host without a GPU, so a plain
    ``pytest tests`` is green there too."""
    try:
        import torch
        have = torch.cuda.is_available()
    except Exception:
        have = False
    exp_lib = "libbetapose_hip_exp" in os.environ.get("BP_LIB", "")
    skip = pytest.mark.skip(reason="no GPU")
    skip_exp = pytest.mark.skip(reason="experimental kernels: needs a GPU and BP_LIB=<libbetapose_hip_exp.so>")
    for it in items:
        if "experimental" in it.keywords and not (have and exp_lib):
            it.add_marker(skip_exp)
        elif "gpu" in it.keywords and not have:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
