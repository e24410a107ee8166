import os
import sys

import pytest

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
sys.path.insert(0, os.path.join(REPO, 'tests', 'golden'))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'slow: long-running CPU emulation case (set STRIVE_SLOW=1)')


def pytest_collection_modifyitems(config, items):
    import torch
    if not torch.cuda.is_available():
        # a plain `pytest tests` on a box without a GPU: the gpu-marked tests are skipped, not failed
        no_gpu = pytest.mark.skip(reason='needs the MI355X (run with -m gpu on the GPU box)')
        for it in items:
            if 'gpu' in it.keywords:
                it.add_marker(no_gpu)
    if os.environ.get('STRIVE_SLOW') == '1':
        return
    skip = pytest.mark.skip(reason='slow emulation case; set STRIVE_SLOW=1')
    for it in items:
        if 'slow' in it.keywords:
            it.add_marker(skip)
