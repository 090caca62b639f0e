import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tip_amd import ops
from tip_amd.data import build_data_dict
from tip_amd.plan import build_rel_plan
dd = build_data_dict(); dev = 'cuda:0'
ei = dd['dd_train_idx'].to(dev); rg = dd['dd_train_range']; R = dd['n_dd_et']; N = 645
rel = torch.repeat_interleave(torch.arange(R), rg[:, 1] - rg[:, 0]).to(dev)
pf = build_rel_plan(ei[1], ei[0], rel, N, R, 256); pb = build_rel_plan(ei[0], ei[1], rel, N, R, 256, backward=True)
def t(f, n=20):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for d in (32, 16):
    y = torch.randn(R * N, d, device=dev); g = torch.randn(N, d, device=dev)
    for dbg in (0, 1, 2, 4, 7, 3):
        os.environ['TIPK_RG_DEBUG'] = str(dbg)
        print('d=%d dbg=%d  fwd %.1f us   bwd %.1f us' % (d, dbg, t(lambda: ops.rel_gather(pf, y, False)), t(lambda: ops.rel_gather(pb, g, True))))
