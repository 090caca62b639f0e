"""A few rel_gather launches on the BioSNAP D-D graph for PMC collection."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tip_amd import ops
from tip_amd.data import build_data_dict
from tip_amd.plan import build_rel_plan
dd = build_data_dict(); dev = 'cuda:0'
ei = dd['dd_train_idx'].to(dev); rg = dd['dd_train_range']; R = dd['n_dd_et']; N = 645
rel = torch.repeat_interleave(torch.arange(R), rg[:, 1] - rg[:, 0]).to(dev)
for d in (32, 16):
    split = ops.rel_gather_split(N, d, False)
    pf = build_rel_plan(ei[1], ei[0], rel, N, R, 256 // split)
    pb = build_rel_plan(ei[0], ei[1], rel, N, R, 256, backward=True)
    y = torch.randn(R * N, d, device=dev); g = torch.randn(N, d, device=dev)
    for _ in range(3):
        ops.rel_gather(pf, y, False, reduce=False)
        ops.rel_gather(pb, g, True)
torch.cuda.synchronize()
