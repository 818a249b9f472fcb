import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

import asac_amd  # noqa: E402,F401  (puts the `algorithm` package on sys.path)

GOLDEN = ROOT / 'tests' / 'golden'


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'slow: long-running variant, additionally gated by an environment variable')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
