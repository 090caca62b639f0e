"""Decompose tipk_rgcn_dy_products (TIPK_DP_DEBUG bits: 1 no loads after the first tile, 2 no second
product / reduction, 4 no first product) at BioSNAP sizes; kernel-only time via a captured graph."""
import os, sys, torch, ctypes as C
sys.path.insert(0, '.')

from tip_amd._lib import lib, ptr, stream_ptr, check, set_option   # needs TIPK_LIB=tip_amd/libtipk_debug.so (make debug)
dev = torch.device('cuda:0')
R, B = 1097, 32
for ncol in (20640, 10320):
    att = torch.randn(R, B, device=dev); xb2 = torch.randn(B, ncol, device=dev); gy = torch.randn(R, ncol, device=dev)
    s_c, s_r = C.c_int(0), C.c_int(0)
    lib().tipk_rgcn_dy_products_plan(R, ncol, B, C.byref(s_c), C.byref(s_r))
    dxb = torch.empty(s_r.value, B, ncol, device=dev); datt = torch.empty(s_c.value, R, B, device=dev)
    def run():
        check(lib().tipk_rgcn_dy_products(ptr(gy), ncol, ptr(att), B, ptr(xb2), ncol, R, ncol, B, None, 0, ptr(dxb), ptr(datt),
                                          stream_ptr(dev)), 'dy')
    for dbg in (0, 1, 2, 4, 3, 6, 7):
        set_option('dp_debug', dbg)
        for _ in range(3): run()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20): run()
        g.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): g.replay()
        b.record(); torch.cuda.synchronize()
        print('ncol %d slabs %d/%d dbg=%d  %.1f us' % (ncol, s_c.value, s_r.value, dbg, a.elapsed_time(b) / 100 * 1e3))
