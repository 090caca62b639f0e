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


def start_ranks(procs):
    """Start spawned rank processes with ONE host thread each, as torch.distributed.run sets it: `world` OpenMP pools
    spinning on the test box's few cores made build_data_dict alone take 144 s in a 4-rank test (2 s in a process of
    its own)."""
    omp = os.environ.get('OMP_NUM_THREADS')
    os.environ['OMP_NUM_THREADS'] = '1'
    try:
        for p in procs:
            p.start()
    finally:
        if omp is None:
            del os.environ['OMP_NUM_THREADS']
        else:
            os.environ['OMP_NUM_THREADS'] = omp


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
