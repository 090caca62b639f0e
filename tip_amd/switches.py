"""Test / experiment switches of the Python op layer, in ONE place.

The library itself reads no environment variable (include/tipk.h: `tipk_set_option`).  The Python side has a handful of route
switches that tests and tools flip to reach the fallback routes on shapes that would otherwise take the fast one.  They are
declared here -- name, what they do -- and read through `on(name)`: an explicit `set(name, value)` wins, else the
environment variable of the same name (read at call time, so `monkeypatch.setenv` in a test works).  Nothing else in the
package reads `os.environ` for dispatch.
"""
import os

SWITCHES = {
    'TIPK_NO_ENCODER_STEP': 'FMEncoder.forward on the per-layer autograd nodes instead of tip_amd/encoder.py',
    'TIPK_NO_RELLOCAL': 'no LDS-resident D-D kernels (rel_gather / stream_gather): the generic gather route',
    'TIPK_NO_DY_FUSED': 'the two products of dY as two grouped GEMMs instead of tipk_rgcn_dy_products',
    'TIPK_NO_CELLS_TWO': 'the pair cells of the two R-GCN layers in two launches',
    'TIPK_NO_DEST_FWD': 'large graphs: the forward pass through Y instead of tipk_rgcn_dest_products',
    'TIPK_NO_ROW_PRODUCTS': 'large graphs: no tipk_rgcn_row_products',
    'TIPK_NO_ROW_PRODUCTS_S': 'large graphs: no tipk_rgcn_row_products_s (wave-uniform form)',
    'TIPK_NO_LAYER_HANDOVER': 'the slab sum between the R-GCN layers as a launch of its own',
    'TIPK_NO_PAIR_BWD': 'the round-4 backward route (compact dY + node_products) behind a pair-form forward pass',
    'TIPK_NO_SYMMETRIC_POS': 'the fused objective evaluates every directed positive (no weight-2 halves)',
    'TIPK_FLOAT_ATOMICS': 'the fused objective on float atomics (not bitwise reproducible)',
    'TIPK_NO_BITMAP': 'the sampler on the binary-search kernel instead of the LDS bitmap',
    'TIPK_NO_PAIR_PRODUCT': 'the pair product on the tiled GEMM instead of tipk_pair_product',
}
_forced = {}


def on(name):
    """True if the switch is set (explicitly, or as a non-empty environment variable)."""
    assert name in SWITCHES, name
    if name in _forced:
        return bool(_forced[name])
    return bool(os.environ.get(name))


def set(name, value):                                          # noqa: A001 (mirrors os.environ semantics: None = back to the environment)
    assert name in SWITCHES, name
    if value is None:
        _forced.pop(name, None)
    else:
        _forced[name] = bool(value)
