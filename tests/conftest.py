import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'reference: needs /root/reference (build container only)')


def pytest_collection_modifyitems(config, items):
    have_ref = os.path.isdir('/root/reference/src')
    skip_ref = pytest.mark.skip(reason='/root/reference not present (GPU box)')
    for item in items:
        if 'reference' in item.keywords and not have_ref:
            item.add_marker(skip_ref)


@pytest.fixture(scope='session')
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


_CKPT_CACHE = {}


def _l3c_checkpoint(calibrated):
    if calibrated not in _CKPT_CACHE:
        import l3c_pytorch_amd  # noqa: F401
        from l3c_pytorch_amd.helpers import config_parser, synthetic
        cfg = config_parser.parse_builtin('ms', 'cr')
        _CKPT_CACHE[calibrated] = (cfg, synthetic.make_state_dict(cfg, 0, calibrated=calibrated))
    return _CKPT_CACHE[calibrated]


@pytest.fixture(scope='session')
def synthetic_l3c():
    """(config_ms, state_dict) of the seeded synthetic L3C checkpoint used by tests/golden/make_golden.py."""
    return _l3c_checkpoint(False)


@pytest.fixture(scope='session')
def synthetic_l3c_cal():
    """(config_ms, state_dict) of the CALIBRATED synthetic checkpoint (tests/golden/make_calibrated.py): mixtures that cover the
    data, every bottleneck level in use."""
    return _l3c_checkpoint(True)


# (fixture file, uses the calibrated checkpoint): every reference-generated L3C network fixture
NET_FIXTURES = [('net_32.npz', False), ('net_cal_32.npz', True), ('net_cal_64x96.npz', True)]


@pytest.fixture(scope='session')
def l3c_checkpoint():
    return _l3c_checkpoint
