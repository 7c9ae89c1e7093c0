import os
import os.path as op
import sys

import pytest

ROOT = op.dirname(op.dirname(op.abspath(__file__)))
for p in (ROOT, op.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'slow: long-running CPU test (exhaustive sweeps)')


@pytest.fixture(scope='session')
def golden_chunks():
    import json
    with open(op.join(ROOT, 'tests', 'golden', 'chunk_cases.json')) as f:
        return json.load(f)


@pytest.fixture(scope='session')
def golden_offsets():
    import json
    with open(op.join(ROOT, 'tests', 'golden', 'offset_cases.json')) as f:
        return json.load(f)
