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
