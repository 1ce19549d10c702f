import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # tests/test_gpu_device_entry.py keeps its device buffers in torch tensors.  torch ships its own HIP runtime; when both
    # it and libzkp_mi355x.so (linked against /opt/rocm) live in one process, torch's must initialise FIRST (the other
    # order leaves the later one without devices), exactly as bench.py does -- so for `-m gpu` sessions on a GPU box it
    # is initialised here, before any test module is imported.
    expr = config.getoption("markexpr", "") or ""
    if "gpu" in expr and "not gpu" not in expr and os.path.exists("/dev/kfd"):
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:
            pass


import pytest  # noqa: E402


@pytest.fixture(autouse=True)
def _gpu_tests_use_the_gpu(request):
    """The host toolbox serves calls of a handful of terms on the host backend (zkp_toolbox_set_host_max_terms, default 16) even when it
    has a GPU context.  The -m gpu tests are there to exercise the DEVICE path at every size, tiny ones included, so they switch that
    off; tests/test_host_backend.py covers the host backend (and, on the GPU box, host == device)."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    from zkp_amd import toolbox as T
    old = T.get_host_max_terms()
    T.set_host_max_terms(0)
    yield
    T.set_host_max_terms(old)
