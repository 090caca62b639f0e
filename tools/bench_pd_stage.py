"""The fused launches of the P -> D stage at BioSNAP size, graph-timed; with the debug build (TIPK_LIB=tip_amd/libtipk_debug.so)
the parts of the backward launch:   python tools/bench_pd_stage.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from tip_amd import _lib, encoder
from tip_amd.data import build_data_dict
from tip_amd.layers import hier_graph
dev = torch.device('cuda:0')
dd = build_data_dict()
dp, npr, n = dd['dp_edge_index'].to(dev), dd['n_prot'], dd['n_drug']
rows = torch.unique(dp[0])
inv = torch.full((npr,), -1, device=dev); inv[rows] = torch.arange(rows.numel(), device=dev)
ns = int(rows.numel())
g = hier_graph(torch.stack([inv[dp[0]], dp[1] - npr + ns]), ns + n, ns, table_rows=ns, d=16)
p, q, ne, c1, nb, d1 = 16, 16, 48, 32, 32, 32
xd, h, w = torch.randn(n, ne, device=dev), torch.randn(ns, p, device=dev), torch.randn(p, q, device=dev)
dn = torch.ones(n, device=dev)
basis, root = torch.randn(nb, ne + q, d1, device=dev), torch.randn(ne + q, d1, device=dev)
xb = torch.zeros(648, nb, 32, device=dev)
f = lambda: encoder.drug_mix_gather_xb(xd, h, w, dn, True, g, basis, root, xb[:, :, :d1])
x0, mean, _ = f()
print('drug_mix_gather_xb_fwd %.1f us' % bench.time_launch_us(f))
gup, agg, w2 = torch.randn(n, ne + q, device=dev), torch.randn(ns, c1, device=dev), torch.randn(c1, p, device=dev).t()
b = lambda: encoder.pd_stage_bwd(gup, dn, mean, w, ne, True, g, agg, w2, None)
print('pd_stage_bwd %.1f us' % bench.time_launch_us(b))
if '+debug' in _lib.build_id():
    for bits, what in ((16, 'up to the edge ids of the first chunk'), (32, 'up to the weighted rows in LDS'), (64, 'up to the row sums of the first chunk'), (1, 'without d W_h'), (2, 'without the gather and what follows'), (3, 'prelude + d xd only'), (4, 'without the d W2 partials'),
                       (5, 'without d W_h and the d W2 partials')):
        _lib.set_option('dm_debug', bits)
        print('   %-44s %.1f us' % (what, bench.time_launch_us(b)))
    _lib.set_option('dm_debug', 0)
