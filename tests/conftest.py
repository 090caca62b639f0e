"""pytest configuration: the `gpu` marker, repo root on sys.path, golden-fixture loader."""
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
    # the library is git-ignored: a clean checkout (or one whose kernels were edited) builds it here,
    # before any test -- and before anything touches the GPU (child `make`, no exec)
    from tip_amd import _lib
    _lib.ensure_built()


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


def load_golden(name, dtype=None):
    """tests/golden/<name>.npz as a dict of torch tensors (floats optionally cast)."""
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    out = {}
    for k in z.files:
        v = z[k]
        if v.dtype.kind in 'US' or v.ndim == 0 and v.dtype.kind in 'iu':
            out[k] = v.item() if v.ndim == 0 else v
            continue
        t = torch.from_numpy(v)
        if dtype is not None and t.is_floating_point():
            t = t.to(dtype)
        out[k] = t
    return out


@pytest.fixture
def golden():
    return load_golden
