"""pytest config: registers the ``gpu`` marker, puts the repo root on
sys.path, exposes the reference-style config paths shipped with the package."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIG_DIR = os.path.join(ROOT, 'sup3r_amd', 'configs')


def pytest_configure(config):
    config.addinivalue_line(
        'markers', 'gpu: test needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def config_dir():
    return CONFIG_DIR
