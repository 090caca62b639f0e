"""Marginal cost of one kind of launch INSIDE the replayed step: DUP=<tip_amd.ops function> [DUP_N=k] runs bench.py
with every call of that function issued k extra times (results unchanged).  (step time with - without) / extra launches =
what such a launch really costs in the graph; rocprof kernel durations overstate small kernels (4.8 us traced, 2.5 us marginal)."""
import os, sys, runpy
sys.path.insert(0, os.getcwd())
from tip_amd import ops
name = os.environ.get('DUP')
if name:
    orig = getattr(ops, name)
    n_extra = int(os.environ.get('DUP_N', '1'))
    def twice(*a, **k):
        for _ in range(n_extra):
            orig(*a, **k)
        return orig(*a, **k)
    setattr(ops, name, twice)
sys.argv = ['bench.py', '--no-cpu-baseline', '--no-kernel-table', '--steps', '50', '--warmup', '10']
runpy.run_path('bench.py', run_name='__main__')
