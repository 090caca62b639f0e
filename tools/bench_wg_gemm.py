"""Time the products that end an R-GCN layer's backward pass (graph-timed launches): one workgroup per output tile
(`wg_gemm_group`) against the split-K slabs + grouped slab sum they replaced, whole group and member by member:
   python tools/bench_wg_gemm.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from tip_amd import ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
for n, d_in, d_out, nb, r, gs in ((645, 64, 32, 32, 1097, 14), (645, 32, 16, 32, 1097, 14)):
    x, g, g_xb = torch.randn(n, d_in, device=dev), torch.randn(n, d_out, device=dev), torch.randn(nb, n, d_out, device=dev)
    basis, root = torch.randn(nb, d_in, d_out, device=dev), torch.randn(d_in, d_out, device=dev)
    slabs = torch.randn(gs, r, nb, device=dev)

    def members():
        return [ops.wg_gemm_job(x.t(), g_xb), ops.wg_gemm_job(x.t(), g),
                ops.wg_gemm_job(g_xb, basis.transpose(1, 2), reduce_batch=True, a2=g, b2=root.t(), gate=x)]

    def old():
        j_basis, j_root = ops.gemm_job(x.t(), g_xb), ops.gemm_job(x.t(), g)
        j_xr = ops.gemm_job(g, root.t(), ksplit=1)
        j_xq = ops.gemm_job(g_xb, basis.transpose(1, 2), out=j_xr.out, c_in=j_xr.out, reduce_batch=True)
        j_xq.gate = x
        ops.gemm_group([j_basis, j_root, j_xr, j_xq], [ops.slab_job(slabs)])
    print('in %d out %d: split-K group + slab sum %.1f us' % (d_in, d_out, bench.time_launch_us(old)))
    print('   wg group (3 products + slab sum) %.1f us' % bench.time_launch_us(lambda: ops.wg_gemm_group(members(), [ops.slab_job(slabs)])))
    for i, what in enumerate(('d basis', 'd root', 'dX')):
        print('   %-8s alone %.1f us' % (what, bench.time_launch_us(lambda: ops.wg_gemm_group([members()[i]]))))
    print('   slab sum alone %.1f us' % bench.time_launch_us(lambda: ops.wg_gemm_group([], [ops.slab_job(slabs)])))
