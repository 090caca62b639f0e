"""Decompose tipk_rel_gather time on a UNIFORM synthetic graph (every relation the same size, so that
TIPK_RG_DEBUG=2 -- no staging after the first unit -- still does the same amount of compute):
dbg 0 = all, 1 = staging only, 2 = compute only."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tip_amd import ops, _lib
from tip_amd.plan import build_rel_plan
dev = 'cuda:0'
N, R, per = 645, 1097, 7590
g = torch.Generator().manual_seed(1)
rel = torch.repeat_interleave(torch.arange(R), per).to(dev)
src = torch.randint(0, N, (R * per,), generator=g).to(dev)
dst = torch.randint(0, N, (R * per,), generator=g).to(dev)


def t(f, n=30):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6


for d in (32, 16):
    split = ops.rel_gather_split(N, d, False)
    pf = build_rel_plan(dst, src, rel, N, R, 256 // split, max_unit=10 ** 9)
    pb = build_rel_plan(src, dst, rel, N, R, 256, backward=True, max_unit=10 ** 9)
    y = torch.randn(R * N, d, device=dev); gg = torch.randn(N, d, device=dev)
    for dbg in (0, 1, 2, 3):
        _lib.set_option('rg_debug', dbg)
        print('uniform d=%d dbg=%d  fwd %.1f us   bwd %.1f us' % (
            d, dbg, t(lambda: ops.rel_gather(pf, y, False, reduce=False)), t(lambda: ops.rel_gather(pb, gg, True))))
