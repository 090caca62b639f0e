"""tip_amd -- the TIP (Tri-graph Information Propagation) training hot path on AMD MI355X.

One path only (BASELINE.json `north_star`): P-P GCN -> P->D mean -> per-relation D-D R-GCN encoder,
DistMult decoder, typed negative sampling -- hand-written gfx950 HIP kernels (`tip_amd/csrc`,
C ABI in `include/tipk.h`) behind the reference's `nn.Module` surface (`tip_amd.layers`).
"""
from .layers import (GCNConv, MyRGCNConv, MyRGCNConv2, MyHierarchyConv, PPEncoder, FMEncoder,  # noqa: F401
                     FMEncoderCat, MultiInnerProductDecoder, NNDecoder, Setting, TIP)
from .neg_sampling import typed_negative_sampling, negative_sampling, manual_seed          # noqa: F401

__version__ = '0.1.0'
