"""A few eager launches of the step's node_products / stream_gather (backward) kernels for PMC collection:
   rocprofv3 --kernel-trace --pmc ... -- python3 tools/pmc_node_products.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tip_amd import ops
from tip_amd.data import build_data_dict
from tip_amd.layers import TIP, Setting
dev = torch.device('cuda:0')
model = TIP(Setting(), dev, data=build_data_dict())
enc = model.encoder
for layer in (enc.rgcn1, enc.rgcn2):
    graph = layer._cache.value
    rs = graph.rs_bwd
    d, nb, r, n = layer.out_channels, layer.num_bases, layer.num_relations, graph.scale.numel()
    g = torch.randn(n, d, device=dev)
    att = torch.randn(r, nb, device=dev)
    # the operands as the step hands them over: XB node-major (rows padded to 32 columns) through strides, and base-innermost
    xb = torch.randn(n, nb, 32 if d <= 32 else d, device=dev).permute(1, 0, 2)[:, :, :d]
    xbt = xb.permute(1, 2, 0).contiguous()
    for _ in range(3):
        dyc = ops.rel_stream_bwd(rs, g, row_scale=graph.scale)
        ops.node_products(dyc, rs.compact, att, xb, xbt)
torch.cuda.synchronize()
