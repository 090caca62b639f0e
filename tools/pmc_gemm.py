"""A few launches of the three large R-GCN products for PMC collection:
rocprofv3 --kernel-trace --pmc <counters> -- python3 tools/pmc_gemm.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tip_amd import ops
dev = 'cuda:0'
R, B, ncol = 1097, 32, 20640
att = torch.randn(R, B, device=dev); xb2 = torch.randn(B, ncol, device=dev); gy = torch.randn(R, ncol, device=dev)
for _ in range(3):
    ops.gemm(att, xb2)
    ops.gemm(att.t(), gy, ksplit=4)
    ops.gemm(gy, xb2.t(), ksplit=57)
torch.cuda.synchronize()
