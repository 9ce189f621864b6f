import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')
    config.addinivalue_line('markers', 'timeout: per-test limit (pytest-timeout)')
    import torch
    # the oracle's CPU ops: a bounded thread pool (the build container has 8 cores and the gloo test spawns 2 more
    # processes; the GPU box has hundreds of logical CPUs and the oracle is what the `-m gpu` suite waits for)
    torch.set_num_threads(min(32 if torch.cuda.is_available() else 4, os.cpu_count() or 1))


@pytest.fixture(scope='session')
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')
