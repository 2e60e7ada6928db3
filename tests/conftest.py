import os
import sys

import pytest

# (before the first torch.cuda call of the session: see vln-ver_amd/__init__.py -- hipGraph replay and memset nodes on ROCm 7.2)
os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)
