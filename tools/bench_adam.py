"""Time tip_amd.optim.Adam's launch on TIP-cat's parameter list (graph-timed):  python tools/bench_adam.py
   (TIPK_LIB=tip_amd/libtipk_debug.so: decomposition through the dm_debug option)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from tip_amd import _lib
from tip_amd.layers import TIP, Setting
from tip_amd.optim import Adam
dev = torch.device('cuda:0')
model = TIP(Setting(), dev)
ps = list(model.parameters())
print('%d tensors, %d floats: %s' % (len(ps), sum(p.numel() for p in ps), [tuple(p.shape) for p in ps]))
for p in ps:
    p.grad = torch.randn_like(p)
opt = Adam(ps, lr=0.01)
opt.step()
print('tipk_adam_step          %.1f us' % bench.time_launch_us(opt.step))
ref = torch.optim.Adam(ps, lr=0.01, capturable=True, fused=True)
ref.step()
print('torch fused capturable  %.1f us' % bench.time_launch_us(ref.step))
if '+debug' in _lib.build_id():
    for dbg, what in ((1, 'no ticket'), (2, 'no bias-correction arithmetic'), (3, 'neither'), (4, 'no stores'), (7, 'loads only')):
        _lib.set_option('dm_debug', dbg)
        print('   %-30s %.1f us' % (what, bench.time_launch_us(opt.step)))
    _lib.set_option('dm_debug', 0)
