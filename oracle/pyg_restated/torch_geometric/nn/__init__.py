"""Restated subset of torch_geometric.nn (PyG 2.0.1) -- TEST INFRASTRUCTURE ONLY."""
from . import conv, models  # noqa: F401
