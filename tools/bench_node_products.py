"""Time the transposed D-D gather and both products of dY on the step's own BioSNAP plans (graph-timed launches):
   python tools/bench_node_products.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from tip_amd import ops, _lib
from tip_amd.data import build_data_dict
from tip_amd.layers import TIP, Setting
dev = torch.device('cuda:0')
model = TIP(Setting(), dev, data=build_data_dict())
enc = model.encoder
for layer in (enc.rgcn1, enc.rgcn2):
    graph = layer._cache.value
    rs = graph.rs_bwd
    d, nb, r, n = layer.out_channels, layer.num_bases, layer.num_relations, graph.scale.numel()
    g = torch.randn(n, d, device=dev)
    att = torch.randn(r, nb, device=dev)
    xb = torch.randn(nb, n, d, device=dev)
    t_g = bench.time_launch_us(lambda: ops.rel_stream_bwd(rs, g, row_scale=graph.scale))
    if rs.compact is not None:
        dyc = ops.rel_stream_bwd(rs, g, row_scale=graph.scale)
        xbt = xb.permute(1, 2, 0).contiguous() if not os.environ.get('NO_XBT') else None
        t_p = bench.time_launch_us(lambda: ops.node_products(dyc, rs.compact, att, xb))
        t_t = bench.time_launch_us(lambda: ops.node_products(dyc, rs.compact, att, xb, xbt))
        xnm = torch.randn(n, nb, 32, device=dev)                 # the step's layout: node-major, rows padded to 32 columns
        t_n = bench.time_launch_us(lambda: ops.node_products(dyc, rs.compact, att, xnm.permute(1, 0, 2)[:, :, :d]))
        print('   node-major XB (the step) %.1f us, + XB^T [N, d, bases] for the d att product %.1f us' % (t_n, t_t))
        if '+debug' in _lib.build_id():                       # TIPK_LIB=tip_amd/libtipk_debug.so: the two roles alone
            for dbg, what in ((16, 'role 1 (dXB) alone'), (8, 'role 2 (d att) alone'), (24, 'empty launch'), (8 + 32, 'role 2, no B loads'), (8 + 64, 'role 2, no A loads'), (8 + 96, 'role 2, pos loads only'),
                              (16 + 128, 'role 1, no att loads'), (16 + 256, 'role 1, no dY loads'), (16 + 384, 'role 1, rel loads only')):
                _lib.set_option('dp_debug', dbg)
                print('   %-22s %.1f us' % (what, bench.time_launch_us(lambda: ops.node_products(dyc, rs.compact, att, xb, xbt))))
            _lib.set_option('dp_debug', 0)
        print('d=%d  rows %d (%.0f %% of R N)  gather %.1f us  node_products %.1f us' % (d, rs.compact.n_rows, 100.0 * rs.compact.n_rows / (r * n), t_g, t_p))
    else:
        dy = ops.rel_stream_bwd(rs, g, row_scale=graph.scale, write_zeros=False).view(r, n * d)
        t_p = bench.time_launch_us(lambda: ops.dy_products(dy, att, xb.view(nb, n * d), rs.row_used, n))
        print('d=%d  dense  gather %.1f us  dy_products (+ slab sums) %.1f us' % (d, t_g, t_p))
