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


def _install_error_log(path):
    """TIPK_ERRLOG=<file>: every assert_close / allclose call appends (call site, max-norm relative error, the fraction of
    its tolerance it used) -- how the tolerances in these tests were calibrated (round 6): each is <= ~20 x the error seen."""
    import json
    import traceback

    def site():
        out = []
        for fr in traceback.extract_stack()[:-2]:
            if os.sep + 'tests' + os.sep in fr.filename and not fr.filename.endswith('conftest.py'):
                out.append('%s:%d' % (os.path.basename(fr.filename), fr.lineno))
        return out[-3:]

    def record(kind, got, want, rtol, atol):
        try:
            g = torch.as_tensor(np.asarray(got.detach().cpu()) if torch.is_tensor(got) else np.asarray(got)).double()
            w = torch.as_tensor(np.asarray(want.detach().cpu()) if torch.is_tensor(want) else np.asarray(want)).double()
            if g.shape != w.shape or g.numel() == 0:
                return
            d = (g - w).abs()
            wmax = float(w.abs().max())
            used = float((d / (atol + rtol * w.abs()).clamp(min=1e-300)).max()) if (atol or rtol) else float(d.max() > 0)
            with open(path, 'a') as f:
                f.write(json.dumps({'site': site(), 'kind': kind, 'n': g.numel(), 'maxnorm_rel': float(d.max()) / max(wmax, 1e-300),
                                    'max_abs': float(d.max()), 'want_max': wmax, 'rtol': rtol, 'atol': atol, 'used': used}) + '\n')
        except Exception:
            pass

    real_close, real_all, real_np = torch.testing.assert_close, torch.allclose, np.testing.assert_allclose

    def assert_close(actual, expected, *a, rtol=None, atol=None, **kw):
        if rtol is not None and atol is not None:
            record('assert_close', actual, expected, rtol, atol)
        return real_close(actual, expected, *a, rtol=rtol, atol=atol, **kw)

    def allclose(a, b, rtol=1e-5, atol=1e-8, **kw):
        record('allclose', a, b, rtol, atol)
        return real_all(a, b, rtol=rtol, atol=atol, **kw)

    def np_close(a, b, rtol=1e-7, atol=0, **kw):
        record('np', a, b, rtol, atol)
        return real_np(a, b, rtol=rtol, atol=atol, **kw)

    torch.testing.assert_close, torch.allclose, np.testing.assert_allclose = assert_close, allclose, np_close


if os.environ.get('TIPK_ERRLOG'):
    _install_error_log(os.environ['TIPK_ERRLOG'])
