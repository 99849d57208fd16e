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
