"""The D-D aggregation launches of one BioSNAP step (+ the pair-form product), a few times each, for PMC collection
(tools/profile_gpu.sh: SQ / LDS counter passes).  Plans and shapes are the step's own (bench.dd_launches)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from tip_amd import ops
from tip_amd.data import build_data_dict
from tip_amd.layers import TIP, Setting
dev = torch.device('cuda:0')
dd = build_data_dict()
model = TIP(Setting(), dev, data=dd)
enc, data = model.encoder, model.data
z = enc(data.d_feat, data.dd_train_idx, data.dd_train_et, data.dd_train_range, data.d_norm, data.p_feat,
        data.pp_train_indices, data.dp_edge_index, data.dp_range_list)          # builds the plans
for rec in bench.dd_launches(enc, dev):
    for _ in range(3):
        rec['fn']()
n, nb = 645, 32
for layer in (enc.rgcn1, enc.rgcn2):
    graph = layer._cache.value
    cells, xb_nb, _zeros = graph.pair_buffers(n, layer.num_bases, layer.out_channels, dev)
    for _ in range(3):
        ops.pair_product(cells, xb_nb, symmetric=graph.pair_fwd.symmetric, links=graph.pair_fwd.links, zeros=_zeros)
torch.cuda.synchronize()
