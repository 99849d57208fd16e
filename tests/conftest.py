import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'slow: full-width CPU oracle cases (tens of seconds)')


@pytest.fixture(scope='session', autouse=True)
def _route_options_of_this_test_process():
    """The library reads no environment variable for its numerics-affecting route options (round 5).  A test that wants another route
    re-runs tests in a subprocess with ES_TEST_VOL_OPTIONS="name=value,..." -- a variable of THIS harness, applied here through the
    explicit API (es_vol_set_option) before any plan is built."""
    spec = os.environ.get('ES_TEST_VOL_OPTIONS', '')
    if spec:
        from echoscene_amd import hip
        for kv in spec.split(','):
            k, v = kv.split('=')
            hip.check(hip.lib().es_vol_set_option(k.encode(), int(v)), 'es_vol_set_option')
    yield


def route_options():
    """the library's current route options as a dict"""
    import ctypes
    from echoscene_amd import hip
    buf = ctypes.create_string_buffer(1024)
    hip.lib().es_options_string(buf, 1024)
    return dict(kv.split('=') for kv in buf.value.decode().strip(';').split(';'))


def load_golden(name):
    d = np.load(os.path.join(GOLDEN, name + '.npz'))
    return {k: torch.from_numpy(np.asarray(d[k])) for k in d.files}


@pytest.fixture(scope='session')
def golden():
    return load_golden


def seeded_state_dict(module, prefix, seed=0):
    """state_dict of a parameter-holder tree filled by the seeded rule (same rule the golden
    generator applied to the reference's modules)."""
    from echoscene_amd import synth
    synth.seeded_fill_(module, seed=seed, prefix=prefix)
    return {k: v.detach() for k, v in module.state_dict().items()}
