"""pytest config: the ``gpu`` marker, import paths, golden-fixture loader.

CPU suite  : python -m pytest tests -x -q -m "not gpu"
GPU suite  : python -m pytest tests -x -q -m gpu      (needs one MI355X + the built C-ABI library)
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_PARENT = os.path.join(ROOT, 'pytorch-nmf_amd')
for p in (ROOT, PKG_PARENT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


@pytest.fixture(scope='session')
def golden():
    return load_golden


def rel_err(a, b):
    import torch
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


def record(name, **vals):
    """Append measured values (errors, tolerances) of a GPU test to gpurun_out/parity_measured.jsonl -- the numbers the
    tolerances in the tests are set from; no-op when the directory cannot be written."""
    import json
    try:
        d = os.path.join(ROOT, 'gpurun_out')
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, 'parity_measured.jsonl'), 'a') as f:
            f.write(json.dumps(dict(test=name, **vals)) + '\n')
    except OSError:
        pass
