"""Time the pair-form backward launches on the step's own BioSNAP plans (graph-timed), and -- with the debug library
(TIPK_LIB=tip_amd/libtipk_debug.so after `make -C tip_amd/csrc debug`) -- `pair_grads` with parts of its work skipped:
   python tools/bench_pair_grads.py"""
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
model().backward()                                            # cells and XB of both layers are in the graphs' buffers
torch.cuda.synchronize()
for layer in (enc.rgcn1, enc.rgcn2):
    graph = layer._cache.value
    pb = graph.pair_bwd
    d, nb, n = layer.out_channels, layer.num_bases, graph.scale.numel()
    cells, xb_nb, _ = graph.pair_buffers(n, nb, d, dev)
    g = torch.randn(n, d, device=dev)
    pg, _ = ops.pair_grads(pb, cells, xb_nb, g)
    t_g = bench.time_launch_us(lambda: ops.pair_grads(pb, cells, xb_nb, g))
    t_a = bench.time_launch_us(lambda: ops.pair_att_gather(pb, pg))
    print('d=%d  slots %d  pair_grads %.1f us   pair_att_gather %.1f us' % (d, pb.n_slots, t_g, t_a))
    if '+debug' in _lib.build_id():
        for dbg, what in ((64, 'role 1 alone (dXB per node)'), (128, 'role 2 alone (dC per tile)'), (1, 'no dC stores'), (2, 'no dC product'),
                          (4, 'no dXB product (cells unread)'), (32, 'no cell loads'), (16 + 64, 'role 1, no g loads'),
                          (8 + 64, 'role 1, at most one tile per wave'), (64 + 4 + 48, 'role 1: slot loads + reduce only'),
                          (128 + 3, 'role 2: loads only')):
            _lib.set_option('dp_debug', dbg)
            print('   %-46s %.1f us' % (what, bench.time_launch_us(lambda: ops.pair_grads(pb, cells, xb_nb, g))))
        _lib.set_option('dp_debug', 0)
