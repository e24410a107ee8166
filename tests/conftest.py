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


@pytest.fixture(autouse=True)
def _strive_option_sync(monkeypatch):
    """The library reads no environment variable (ABI 17): its switches are options that the Python host sets from STRIVE_<NAME>
    variables when it loads the library.  Tests flip such variables with monkeypatch in the middle of a process, so every
    monkeypatch.setenv / delenv of a STRIVE_ variable re-synchronises the options of every loaded library (product and emulator),
    and every test starts from the environment as it is."""
    from strive_amd import _lib
    orig_set, orig_del = monkeypatch.setenv, monkeypatch.delenv

    def setenv(name, value, *a, **k):
        orig_set(name, value, *a, **k)
        if name.startswith('STRIVE_'):
            _lib.sync_all_options_from_env()

    def delenv(name, *a, **k):
        orig_del(name, *a, **k)
        if name.startswith('STRIVE_'):
            _lib.sync_all_options_from_env()

    monkeypatch.setenv, monkeypatch.delenv = setenv, delenv
    _lib.sync_all_options_from_env()
    yield
