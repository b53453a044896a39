import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    if os.environ.get('AA_TEST_FILL_NAN') == '1':
        # audit mode: every torch.empty of the run starts as NaN (torch's deterministic mode), so a kernel that reads a row nobody wrote fails its test instead of
        # passing on the allocator's finite leftovers (round 6: a pad row of the window gather did exactly that).  `AA_TEST_FILL_NAN=1 python -m pytest tests -m gpu`
        import torch
        torch.use_deterministic_algorithms(True, warn_only=True)
        torch.utils.deterministic.fill_uninitialized_memory = True


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')
