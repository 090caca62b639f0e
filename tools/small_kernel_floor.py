"""What the SMALL kernels of the step cost when replayed in a hipGraph, alone and alternating
(launch floor on this box: tools/microbench/launch_floor.hip = 1.6 us per dependent node)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tip_amd import ops

dev = 'cuda:0'
torch.manual_seed(0)
x = torch.randn(645, 64, device=dev)
x2 = torch.randn(645, 64, device=dev)
dn = torch.rand(645, device=dev) + 0.5
slabs10 = torch.randn(10, 16, 16, device=dev)
slabs128 = torch.randn(128, 645, 32, device=dev)
mean = torch.randn(645, 16, device=dev)
w = torch.randn(16, 16, device=dev)
xb_in = torch.randn(645, 64, device=dev)
basis = torch.randn(32, 64, 32, device=dev)
root = torch.randn(64, 32, device=dev)
scale = torch.rand(645, device=dev)
addend = torch.randn(645, 32, device=dev)


def timed(name, fn, n=40):
    fn(); torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for _ in range(n):
                fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print('%-64s %6.2f us / call' % (name, e0.elapsed_time(e1) * 1000 / 10 / n), flush=True)


out_ra = torch.empty_like(x)
timed('rows_affine 645x64 /d_norm', lambda: ops.rows_affine(x, row_div=dn, out=out_ra))
timed('sum_slabs 10 x 256', lambda: ops.sum_slabs(slabs10))
timed('sum_slabs 128 x 20640 (+scale, addend, relu)', lambda: ops.sum_slabs(slabs128, row_scale=scale, addend=addend, relu=True))
timed('gemm 645x16x16', lambda: ops.gemm(mean, w))
timed('gemm_group XB (645x64 . 32x64x32) + X root', lambda: ops.gemm_group([ops.gemm_job(xb_in, basis), ops.gemm_job(xb_in, root)]))
timed('gemm 16x16x645 (k-split 10) + slab sum', lambda: ops.gemm(mean.t(), mean))


def alternating():
    ops.rows_affine(x, row_div=dn, out=out_ra)
    ops.sum_slabs(slabs10)
    ops.gemm(mean, w)
timed('alternating rows_affine / sum_slabs 10x256 / gemm 645x16x16 (per 3)', alternating, n=15)


def chain():
    a = ops.rows_affine(x, row_div=dn)
    b = ops.rows_affine(a, row_div=dn)
    c = ops.rows_affine(b, row_div=dn)
    return c
timed('3 dependent rows_affine (per 3)', chain, n=15)

# the same kernels right after a launch that streams 512 MB (what the step's large kernels do to the caches and TLBs)
big_a = torch.empty(64 << 20, device=dev)           # 256 MB
big_b = torch.empty(64 << 20, device=dev)
def thrash():
    big_b.copy_(big_a)
timed('thrash: 256 MB -> 256 MB copy', thrash, n=10)
def t_ra():
    thrash(); ops.rows_affine(x, row_div=dn, out=out_ra)
timed('thrash + rows_affine', t_ra, n=10)
def t_3():
    thrash(); ops.rows_affine(x, row_div=dn, out=out_ra); ops.sum_slabs(slabs10); ops.gemm(mean, w)
timed('thrash + rows_affine + sum_slabs 10x256 + gemm 645x16x16', t_3, n=10)
def t_33():
    thrash()
    for _ in range(3):
        ops.rows_affine(x, row_div=dn, out=out_ra)
timed('thrash + 3 x rows_affine', t_33, n=10)
y32 = torch.randn(1097 * 645, 32, device=dev)
y32b = torch.empty_like(y32)
def thrash2():
    y32b.copy_(y32)
timed('thrash2: 90 MB copy', thrash2, n=10)
def t2_ra():
    thrash2(); ops.rows_affine(x, row_div=dn, out=out_ra)
timed('thrash2 + rows_affine', t2_ra, n=10)
