import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tip_amd import ops
from tip_amd.data import build_data_dict
from tip_amd.neg_sampling import typed_negative_sampling
dd = build_data_dict()
dev = 'cuda:0'
pos = dd['dd_train_idx'].to(dev); et = dd['dd_train_et'].to(dev); rg = dd['dd_train_range'].to(dev)
z = torch.randn(645, 16, device=dev) * 0.5; w = torch.randn(dd['n_dd_et'], 16, device=dev) * 0.3
neg = typed_negative_sampling(pos, 645, rg)
def t(f, n=10):
    """ms per call, n calls captured in one hipGraph and replayed (kernel time back to back: the eager loop is bound by the
    host's launch rate below ~0.18 ms per call)"""
    f(); torch.cuda.synchronize()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        f()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * n)
print('loss+grad ms', t(lambda: ops.distmult_loss(z, w, pos, neg, et)))
print('loss only ms', t(lambda: ops.distmult_loss(z, w, pos, neg, et, need_grad=False)))
print('sampler   ms', t(lambda: typed_negative_sampling(pos, 645, rg)))
print('fwd score ms', t(lambda: ops.distmult_fwd(z, w, pos, et)))
pos32, neg32, et32 = pos.int(), neg.int(), et.int()
ops.relation_tasks(et32)
print('loss+grad int32 ms', t(lambda: ops.distmult_loss(z, w, pos32, neg32, et32)))
negp = typed_negative_sampling(pos, 645, rg, packed=True)
print('loss+grad packed pairs ms %.3f   loss only %.3f   sampler packed %.3f' % (
    t(lambda: ops.distmult_loss(z, w, pos, negp, et)), t(lambda: ops.distmult_loss(z, w, pos, negp, et, need_grad=False)),
    t(lambda: typed_negative_sampling(pos, 645, rg, packed=True))))

from tip_amd import _lib
_lib.set_option('dm_task_kernel', 1)
print('task kernel (k/4 lanes per position): loss+grad ms %.3f   loss only ms %.3f' % (
    t(lambda: ops.distmult_loss(z, w, pos, neg, et)), t(lambda: ops.distmult_loss(z, w, pos, neg, et, need_grad=False))))
_lib.set_option('dm_task_kernel', 0)
if '+debug' in _lib.build_id():                               # TIPK_LIB=tip_amd/libtipk_debug.so
    _lib.set_option('dm_debug', 1)
    print('no fixed-point adds at all: loss+grad ms %.3f' % t(lambda: ops.distmult_loss(z, w, pos, neg, et)))
    _lib.set_option('dm_debug', 4)
    print('no flush of the d z image:  loss+grad ms %.3f' % t(lambda: ops.distmult_loss(z, w, pos, neg, et)))
    _lib.set_option('dm_debug', 0)
for rep in range(3):
    print('again (default options): loss+grad packed pairs ms %.3f   loss only %.3f' % (
        t(lambda: ops.distmult_loss(z, w, pos, negp, et)), t(lambda: ops.distmult_loss(z, w, pos, negp, et, need_grad=False))))
