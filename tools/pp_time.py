import os, sys, torch
sys.path.insert(0, os.getcwd())
from tip_amd import ops
from bench import time_launch_us
dev='cuda:0'
n, nb = 645, 32
for d in (32, 16):
    n_pad = 648
    cells = torch.zeros(n_pad, n, nb, device=dev)
    mask = (torch.rand(n, n, 1, device=dev) < 0.3).float()
    cells[:n] = torch.randn(n, n, nb, device=dev) * mask
    xb = torch.zeros(n_pad, nb, d, device=dev); xb[:n] = torch.randn(n, nb, d, device=dev)
    print('d', d, 'pair_product us', time_launch_us(lambda: ops.pair_product(cells, xb)))
    os.environ['TIPK_NO_PAIR_PRODUCT'] = '1'
    print('d', d, 'tiled gemm   us', time_launch_us(lambda: ops.pair_product(cells, xb)))
    del os.environ['TIPK_NO_PAIR_PRODUCT']
    big = torch.empty(128 << 20, device=dev); big2 = torch.empty(128 << 20, device=dev)   # 512 MB each
    def cold():
        big2.copy_(big)
        ops.pair_product(cells, xb)
    def thrash():
        big2.copy_(big)
    print('d', d, 'cold: thrash+pp', time_launch_us(cold, reps=5), 'thrash', time_launch_us(thrash, reps=5))
    del big, big2
