"""Where does the one-off preprocessing time go on the GPU box?  (plan builds of the first step, library load)"""
import sys, time, torch, cProfile, pstats
sys.path.insert(0, '.')
t00 = time.perf_counter()
from tip_amd import ops, _lib
from tip_amd.data import build_data_dict
from tip_amd import plan as P
from tip_amd.layers import gcn_norm_graph, hier_graph, rgcn_graph, relation_of_edges
dev = torch.device('cuda:0')
def T(name, f):
    torch.cuda.synchronize(); t = time.perf_counter(); r = f(); torch.cuda.synchronize()
    print('%-52s %.3f s' % (name, time.perf_counter() - t)); return r
dd = T('build_data_dict (host)', build_data_dict)
T('lib load + first call', lambda: _lib.lib().tipk_abi_version())
T('cuda init + tiny alloc', lambda: torch.zeros(1, device=dev))
ei = T('edges to device', lambda: dd['dd_train_idx'].to(dev)); rg = dd['dd_train_range']; R = int(dd['n_dd_et']); N = 645
rel = T('relation_of_edges', lambda: relation_of_edges(rg, ei.shape[1], dev))
T('warm-up sort', lambda: torch.sort(rel * N + ei[0]))
src, dst = ei[0], ei[1]
n_cu = 256
T('pair plan (symmetry check + half edges, L=8)', lambda: P.build_stream_plan_rows(src[src <= dst] * N + dst[src <= dst], rel[src <= dst], N * N, R, n_cu, 8, 4))
T('transposed plan d=32 (compact, L=8)', lambda: P.build_stream_plan(src, dst, rel, N, R, n_cu, 8, 4, compact=True))
T('transposed plan d=16 (compact, L=4)', lambda: P.build_stream_plan(src, dst, rel, N, R, n_cu, 4, 4, compact=True))
T('rgcn_graph d=32 (pair + transposed + degree)', lambda: rgcn_graph(ei, rel, N, R, None, d_out=32, n_bases=32))
T('rgcn_graph d=16', lambda: rgcn_graph(ei, rel, N, R, None, d_out=16, n_bases=32))
pp = dd['pp_train_indices'].to(dev); n_prot = int(dd['n_prot'])
T('gcn_norm_graph d=32', lambda: gcn_norm_graph(pp, n_prot, None, 32))
T('gcn_norm_graph d=16', lambda: gcn_norm_graph(pp, n_prot, None, 16))
dp = dd['dp_edge_index'].to(dev)
T('hier_graph', lambda: hier_graph(dp, n_prot + N, n_prot, None, table_rows=n_prot, d=16))
pr = cProfile.Profile()
pr.enable()
P.build_stream_plan(src, dst, rel, N, R, n_cu, 8, 4, compact=True)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
print('total %.2f s' % (time.perf_counter() - t00))
