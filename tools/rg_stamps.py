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
import os as _os
_os.environ['TIPK_NO_RELSTREAM'] = '1'            # the stamps live in the relation-local kernel: build ITS backward plan
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
        tot, loop, commit, wait = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
        print('d=%d %s  workgroups %d x 16 waves' % (d, 'bwd' if bwd else 'fwd', n))
        print('   wave lifetime      mean %8.0f  max %8.0f cycles (100 MHz ticks x?)' % (tot.mean(), tot.max()))
        print('   position loops     mean %8.0f  (%.0f %%)   per-WG max/mean over waves %.2f' % (
            loop.mean(), 100 * loop.mean() / tot.mean(), (loop.max(1) / loop.mean(1)).mean()))
        print('   commit             mean %8.0f  (%.0f %%)' % (commit.mean(), 100 * commit.mean() / tot.mean()))
        print('   wait at barrier    mean %8.0f  (%.0f %%)' % (wait.mean(), 100 * wait.mean() / tot.mean()))
        pro, epi = a[..., 4], a[..., 5]
        print('   start -> 1st unit  mean %8.0f  (%.0f %%)' % (pro.mean(), 100 * pro.mean() / tot.mean()))
        print('   epilogue           mean %8.0f  (%.0f %%)' % (epi.mean(), 100 * epi.mean() / tot.mean()))
        pref, rel_ = a[..., 6], a[..., 7]
        print('   prefetch issue     mean %8.0f  (%.0f %%)' % (pref.mean(), 100 * pref.mean() / tot.mean()))
        print('   id chunk reloads   mean %8.0f  (%.0f %%)' % (rel_.mean(), 100 * rel_.mean() / tot.mean()))
        print('   other              %.0f %%' % (100 * (1 - (loop + commit + wait + pro + epi + pref + rel_).mean() / tot.mean())))
        print('   per-WG lifetime (max wave): min %8.0f  mean %8.0f  max %8.0f' % (tot.max(1).min(), tot.max(1).mean(), tot.max(1).max()))
