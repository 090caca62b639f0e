"""Time tipk_rel_gather on the BioSNAP D-D graph for several work-unit sizes (and debug modes).
usage: python3 tools/bench_relgather.py [max_unit ...]   (0 = one unit per relation)"""
import os, sys, time, torch
# compute-skipping switches exist only in the -DTIPK_DEBUG library: `make -C tip_amd/csrc debug`, TIPK_LIB=tip_amd/libtipk_debug.so
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tip_amd import ops, _lib
from tip_amd.data import build_data_dict
from tip_amd.plan import build_rel_plan
dd = build_data_dict(); dev = 'cuda:0'
ei = dd['dd_train_idx'].to(dev); rg = dd['dd_train_range']; R = dd['n_dd_et']; N = 645
rel = torch.repeat_interleave(torch.arange(R), rg[:, 1] - rg[:, 0]).to(dev)


def t(f, n=30):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6


for mu in [int(x) for x in sys.argv[1:]] or [0]:
    mu_ = mu or 10 ** 9
    for d in (32, 16):
        split = ops.rel_gather_split(N, d, False)
        pf = build_rel_plan(ei[1], ei[0], rel, N, R, 256 // split, max_unit=mu_)
        pb = build_rel_plan(ei[0], ei[1], rel, N, R, 256, backward=True, max_unit=mu_)
        y = torch.randn(R * N, d, device=dev); g = torch.randn(N, d, device=dev)
        for dbg in ((0, 1) if mu == 0 else (0,)):
            _lib.set_option('rg_debug', dbg)
            print('max_unit %6d units %d/%d d=%d dbg=%d  fwd %.1f us   bwd %.1f us' % (
                mu, pf.n_units, pb.n_units, d, dbg, t(lambda: ops.rel_gather(pf, y, False, reduce=False)),
                t(lambda: ops.rel_gather(pb, g, True))))
