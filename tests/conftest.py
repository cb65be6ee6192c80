import os
import sys

import pytest

# before anything touches the GPU: hipGraph replay is only safe with the runtime's graph packet capture off (pixray_amd/__init__.py)
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
# CPU suite on a small shared container: idle OpenMP workers sleep instead of spinning (a spinning worker that loses its core to
# another thread stalls every barrier of a fork-join region: the torch-heavy oracle / StyleLoss tests were seen 60x slower in
# some whole-suite runs and never alone); set before torch loads its OpenMP runtime
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
os.environ.setdefault("GOMP_SPINCOUNT", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")
    if not _have_gpu():
        import torch
        torch.set_num_threads(max(1, min(torch.get_num_threads(), (os.cpu_count() or 4) - 2)))      # leave room for helper threads


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container (gpu tests run through gpurun)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
