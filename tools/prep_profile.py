"""Where does the one-off preprocessing time go on the GPU box?  (plan builds, library load, first launches)"""
import sys, time, torch
sys.path.insert(0, '.')
t00 = time.perf_counter()
from tip_amd import ops, _lib
from tip_amd.data import build_data_dict
from tip_amd import plan as P
from tip_amd.layers import gcn_norm_graph, hier_graph, rgcn_graph, relation_of_edges
dev = torch.device('cuda:0')
def T(name, f):
    torch.cuda.synchronize(); t = time.perf_counter(); r = f(); torch.cuda.synchronize()
    print('%-44s %.3f s' % (name, time.perf_counter() - t)); return r
dd = T('build_data_dict (host)', build_data_dict)
T('lib load + first call', lambda: _lib.lib().tipk_abi_version())
T('cuda init + tiny alloc', lambda: torch.zeros(1, device=dev))
ei = T('edges to device', lambda: dd['dd_train_idx'].to(dev)); rg = dd['dd_train_range']; R = int(dd['n_dd_et']); N = 645
rel = T('relation_of_edges', lambda: relation_of_edges(rg, ei.shape[1], dev))
T('rel plan fwd (128 wg)', lambda: P.build_rel_plan(ei[1], ei[0], rel, N, R, 128))
T('rel plan bwd', lambda: P.build_rel_plan(ei[0], ei[1], rel, N, R, 256, backward=True))
T('rgcn_graph d=32 (both rel plans + degree)', lambda: rgcn_graph(ei, rel, N, R, None, d_out=32))
T('rgcn_graph d=16', lambda: rgcn_graph(ei, rel, N, R, None, d_out=16))
pp = dd['pp_train_indices'].to(dev); n_prot = int(dd['n_prot'])
T('gcn_norm_graph d=32', lambda: gcn_norm_graph(pp, n_prot, None, 32))
T('gcn_norm_graph d=16', lambda: gcn_norm_graph(pp, n_prot, None, 16))
dp = dd['dp_edge_index'].to(dev)
T('hier_graph', lambda: hier_graph(dp, n_prot + N, n_prot, None, table_rows=n_prot, d=16))
x = torch.randn(1097 * 645, 32, device=dev)
g = rgcn_graph(ei, rel, N, R, None, d_out=32)
T('first rel_gather launch', lambda: ops.rel_gather(g.rl_fwd, x, False))
T('second rel_gather launch', lambda: ops.rel_gather(g.rl_fwd, x, False))
print('total %.2f s' % (time.perf_counter() - t00))
