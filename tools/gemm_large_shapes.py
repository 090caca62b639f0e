"""Config-5-shaped dense products (graph-free timing): T . basis with and without slabs of bases, XB, dX, d basis.
   python tools/gemm_large_shapes.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tip_amd import ops
dev='cuda:0'
n, nb, d = 10000, 32, 128
t_b = torch.randn(nb, n, d, device=dev); basis = torch.randn(nb, d, d, device=dev)
def t(f, reps=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
ref = ops.gemm(t_b, basis, reduce_batch=True)
for kg in (None, 16, 8, 4, 2):
    f = (lambda kg=kg: ops.gemm_group([ops.gemm_job(t_b, basis, reduce_batch=True, kgroup=kg)])[0]) if kg else (lambda: ops.gemm(t_b, basis, reduce_batch=True))
    out = f()
    print('T.basis kgroup', kg, '%.1f us' % t(f), float((out - ref).abs().max() / ref.abs().max()))
x = torch.randn(n, d, device=dev)
print('XB  %.1f us' % t(lambda: ops.gemm(x, basis)))
g_xb = torch.randn(nb, n, d, device=dev)
print('dX = sum_b dXB_b basis_b^T  %.1f us' % t(lambda: ops.gemm(g_xb, basis.transpose(1, 2), reduce_batch=True)))
print('d basis = x^T dXB  %.1f us' % t(lambda: ops.gemm(x.t(), g_xb)))
