"""conv2 of the P-P encoder on the rows the P->D stage reads (BioSNAP): transform-first (product + 16-column gather) against
aggregate-first (32-column gather with the dense map in its epilogue), forward launches graph-timed:
   python tools/bench_gcn_rows.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from tip_amd import ops
from tip_amd.data import build_data_dict
from tip_amd.layers import gcn_norm_graph
dev = torch.device('cuda:0')
dd = build_data_dict()
ei = dd['pp_train_indices'].to(dev)
n = dd['n_prot']
rows = torch.unique(dd['dp_edge_index'][0].to(torch.int64)).to(dev)
rows = rows[rows < n]
print('proteins', n, 'kept rows', rows.numel())
x = torch.randn(n, 32, device=dev)
wt = torch.randn(32, 16, device=dev)
w, b = wt.t(), torch.randn(16, device=dev)
g16 = gcn_norm_graph(ei, n, d=16, rows=rows)
g32 = gcn_norm_graph(ei, n, d=32, rows=rows)
t_mm = bench.time_launch_us(lambda: ops.gemm(x, w.t()))
xl = ops.gemm(x, w.t())
t_g16 = bench.time_launch_us(lambda: ops.gather_sum(g16.fwd, xl, bias=b))
t_g32 = bench.time_launch_us(lambda: ops.gather_sum(g32.fwd, x))
t_lin = bench.time_launch_us(lambda: ops.gather_sum_lin(g32.fwd, x, w, b))
print('transform-first: product %.1f us + gather (16 columns) %.1f us;  aggregate-first: gather (32 columns) %.1f us, with the dense map %.1f us'
      % (t_mm, t_g16, t_g32, t_lin))
up = torch.randn(rows.numel(), 16, device=dev)
t_b16 = bench.time_launch_us(lambda: ops.gather_sum(g16.bwd, up))
up32 = torch.randn(rows.numel(), 32, device=dev)
t_b32 = bench.time_launch_us(lambda: ops.gather_sum(g32.bwd, up32))
print('transposed gather: 16 columns %.1f us, 32 columns %.1f us' % (t_b16, t_b32))
