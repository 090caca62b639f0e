"""Where the relation-local gather spends its cycles: per-wave stamps of the -DTIPK_DEBUG library
(`make -C tip_amd/csrc debug`; run with TIPK_LIB=tip_amd/libtipk_debug.so).  Prints, per launch shape,
the split of a workgroup's lifetime into position loops / commit (staging into LDS between barriers) /
waiting at the unit barrier, and the imbalance between the waves of a workgroup and between workgroups."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tip_amd import ops, _lib
from tip_amd.data import build_data_dict
from tip_amd.layers import rgcn_graph
dd = build_data_dict(); dev = 'cuda:0'
ei = dd['dd_train_idx'].to(dev); rg = dd['dd_train_range']; R = dd['n_dd_et']; N = 645
rel = torch.repeat_interleave(torch.arange(R), rg[:, 1] - rg[:, 0]).to(dev)
L = _lib.lib()
buf = (C.c_ulonglong * (512 * 16 * 8))()
for d in (32, 16):
    graph = rgcn_graph(ei, rel, N, R, d_out=d)
    y = torch.randn(R * N, d, device=dev); g = torch.randn(N, d, device=dev)
    for bwd in (False, True):
        rp = graph.rl_bwd if bwd else graph.rl_fwd
        fn = (lambda: ops.rel_gather(rp, g, True, row_scale=graph.scale)) if bwd else (lambda: ops.rel_gather(rp, y, False, reduce=False))
        for _ in range(3): fn()
        torch.cuda.synchronize()
        assert L.tipk_debug_rg_stamps(buf) == 0
        split = ops.rel_gather_split(N, d, bwd)
        n = rp.n_wg * split
        a = np.frombuffer(buf, dtype=np.uint64).reshape(512, 16, 8)[:n].astype(np.float64)
        tot, loop, wait, first, epi, issue, stages = a[..., 0], a[..., 1], a[..., 3], a[..., 4], a[..., 5], a[..., 6], a[..., 7]
        print('d=%d %s  workgroups %d x 16 waves, %.1f stages per workgroup' % (d, 'bwd' if bwd else 'fwd', n, stages.mean()))
        pc = lambda x: '%8.0f (%2.0f %%)' % (x.mean(), 100 * x.mean() / tot.mean())
        print('   wave lifetime        %8.0f cycles (max %.0f)' % (tot.mean(), tot.max()))
        print('   position loops     ', pc(loop), '  per-WG max/mean over waves %.2f' % (loop.max(1) / loop.mean(1)).mean())
        print('   stage barrier      ', pc(wait), '  (wait for the slowest wave + DMA drain)')
        print('   issuing DMA        ', pc(issue))
        print('   start -> 1st stage ', pc(first))
        print('   epilogue           ', pc(epi))
        print('   other               %.0f %%' % (100 * (1 - (loop + wait + issue + first + epi).mean() / tot.mean())))
