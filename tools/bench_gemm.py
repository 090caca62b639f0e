"""Time the three large products of an R-GCN layer (Y = att.XB, dXB = att^T.dY, d att = dY.XB^T) at
BioSNAP sizes for several split factors.  usage: python3 tools/bench_gemm.py [n_cols ...]"""
import sys
import torch
sys.path.insert(0, '.')
from tip_amd import ops, _lib

dev = torch.device('cuda:0')
R, B = 1097, 32


def timeit(fn, iters=20):
    """GPU time per call: the calls are captured into one hipGraph (no host launch cost in the number)."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (5 * iters) * 1e3


for ncol in [int(x) for x in sys.argv[1:]] or [20640, 10320]:
    att = torch.randn(R, B, device=dev)
    xb2 = torch.randn(B, ncol, device=dev)
    gy = torch.randn(R, ncol, device=dev)
    mb = R * ncol * 4 / 1e6
    t = timeit(lambda: ops.gemm(att, xb2))
    print('ncol %d  Y            %7.1f us  %.2f TB/s (write %.0f MB)' % (ncol, t, mb / t, mb))
    for ks in (2, 4, 8, 16):
        t = timeit(lambda: ops.gemm(att.t(), gy, ksplit=ks))
        print('ncol %d  dXB  k%-3d    %7.1f us  %.2f TB/s' % (ncol, ks, t, mb / t))
    for ks in (19, 38, 57, 114, 128):
        t = timeit(lambda: ops.gemm(gy, xb2.t(), ksplit=ks))
        print('ncol %d  dAtt k%-3d    %7.1f us  %.2f TB/s' % (ncol, ks, t, mb / t))
    t = timeit(lambda: ops.dy_products(gy, att, xb2))
    print('ncol %d  dy_products     %7.1f us  %.2f TB/s (one read, incl. slab sums)' % (ncol, t, mb / t))
    t = timeit(lambda: ops.gemm_group([ops.gemm_job(gy, xb2.t()), ops.gemm_job(att.t(), gy)]))
    print('ncol %d  group(dAtt,dXB) %7.1f us  %.2f TB/s (one read)' % (ncol, t, mb / t))
    import os
    _lib.set_option('gemm_no_stream', 1)
    t = timeit(lambda: ops.gemm(att, xb2))
    print('ncol %d  [tiled] Y       %7.1f us  %.2f TB/s' % (ncol, t, mb / t))
    t = timeit(lambda: ops.gemm(att.t(), gy, ksplit=4))
    print('ncol %d  [tiled] dXB k4  %7.1f us  %.2f TB/s' % (ncol, t, mb / t))
    t = timeit(lambda: ops.gemm(gy, xb2.t(), ksplit=57))
    print('ncol %d  [tiled] dAtt k57 %7.1f us  %.2f TB/s' % (ncol, t, mb / t))
    _lib.set_option('gemm_no_stream', 0)
    t = timeit(lambda: gy.copy_(xb2[:1].expand(R, ncol)))
    print('ncol %d  torch fill      %7.1f us  %.2f TB/s' % (ncol, t, mb / t))
    t = timeit(lambda: gy.sum())
    print('ncol %d  torch sum       %7.1f us  %.2f TB/s' % (ncol, t, mb / t))
