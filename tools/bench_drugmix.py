"""Time the fused P -> D + drug-mix launches on the BioSNAP graph, whole and in parts (graph-timed launches):
   python tools/bench_drugmix.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from tip_amd import ops
from tip_amd._lib import lib, check, ptr, stream_ptr
from tip_amd.data import build_data_dict
from tip_amd.layers import hier_graph
dev = torch.device('cuda:0')
dd = build_data_dict()
dp, npr, n = dd['dp_edge_index'].to(dev), dd['n_prot'], dd['n_drug']
rows = torch.unique(dp[0])
inv = torch.full((npr,), -1, device=dev); inv[rows] = torch.arange(rows.numel(), device=dev)
ns = int(rows.numel())
g = hier_graph(torch.stack([inv[dp[0]], dp[1] - npr + ns]), ns + n, ns, table_rows=ns, d=16)
csr = g.pd_csr
for (p, q, ne, cat) in ((16, 16, 48, 1), (16, 64, 64, 0)):
    xd, h, w = torch.randn(n, ne, device=dev), torch.randn(ns, p, device=dev), torch.randn(p, q, device=dev)
    dn = torch.ones(n, device=dev)
    out = torch.empty(n, ne + q if cat else ne, device=dev); mean = torch.empty(n, p, device=dev)
    gup = torch.randn_like(out)
    g_xd, g_h, g_w = torch.empty(n, ne, device=dev), torch.empty(ns, p, device=dev), torch.empty(p, q, device=dev)
    def fwd():
        check(lib().tipk_drug_mix_gather_fwd(ptr(xd), ne, ptr(dn), ptr(h), p, ptr(csr['fwd_ptr']), ptr(csr['fwd_src']), ptr(csr['scale']),
                                             ptr(csr['fwd_wg']), ptr(csr['fwd_order']), csr['fwd_wg'].shape[0], ptr(w), p, q, n, ne, cat, ptr(out), out.stride(0),
                                             ptr(mean), stream_ptr(dev)), 'fwd')
    g_mean = torch.empty(n, p, device=dev)
    def bwd(want_xd=True, want_m=True):
        check(lib().tipk_drug_mix_bwd(ptr(gup), gup.stride(0), ptr(dn), ptr(mean), ptr(w), p, q, n, ne, cat, ptr(g_xd) if want_xd else None,
                                      ne, ptr(g_mean) if want_m else None, ptr(g_w), stream_ptr(dev)), 'bwd')
    fwd()
    print('p %d q %d cat %d: fwd %.1f us  bwd %.1f us  (d W only %.1f)  transposed gather %.1f us' % (
        p, q, cat, bench.time_launch_us(fwd), bench.time_launch_us(bwd), bench.time_launch_us(lambda: bwd(False, False)),
        bench.time_launch_us(lambda: ops.gather_sum(g.bwd, g_mean))))
