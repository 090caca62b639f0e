"""One fused-objective launch for PMC collection (rocprofv3 --pmc ... -- python3 tools/pmc_decoder.py [nograd])."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tip_amd import ops
from tip_amd.data import build_data_dict
from tip_amd.neg_sampling import typed_negative_sampling
dd = build_data_dict(); dev = 'cuda:0'
pos = dd['dd_train_idx'].to(dev); et = dd['dd_train_et'].to(dev); rg = dd['dd_train_range'].to(dev)
z = torch.randn(645, 16, device=dev) * 0.5; w = torch.randn(dd['n_dd_et'], 16, device=dev) * 0.3
neg = typed_negative_sampling(pos, 645, rg, packed='plain' not in sys.argv)     # the training step's form of the negatives
for _ in range(3):
    ops.distmult_loss(z, w, pos, neg, et, need_grad='nograd' not in sys.argv)
torch.cuda.synchronize()
