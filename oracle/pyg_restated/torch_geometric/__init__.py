"""TEST INFRASTRUCTURE ONLY -- restatement of the four PyTorch-Geometric 2.0.1 symbols the
reference's hot path imports (`/root/reference/src/layers.py:2,6,8`):

    torch_geometric.nn.conv.MessagePassing / GCNConv
    torch_geometric.nn.models.InnerProductDecoder
    torch_geometric.data.Data

PyG 2.0.1 / torch-scatter 2.0.8 (`/root/reference/environment_tip_gpu.yml:69,79`) are third-party
dependencies that are NOT vendored under /root/reference and are not installed in this image (no
network).  This package restates their *published* semantics in plain torch so that the reference's
own `src/layers.py` can be imported UNCHANGED in the build container to generate golden vectors
(`oracle/make_golden.py`).  It never travels into the product: nothing under `tip_amd/` imports it.

Parity status: the reference holds no tests for this boundary, so the PyG part of the oracle is
"unpinned" (see DESIGN.md section "Oracle").  Everything above PyG (the reference's own classes) is
executed literally.
"""
__version__ = "2.0.1-restated"
