"""Time `pair_product` on the step's own BioSNAP buffers (graph-timed), and -- with the debug library
(TIPK_LIB=tip_amd/libtipk_debug.so after `make -C tip_amd/csrc debug`) -- with parts of its work skipped:
   python tools/bench_pair_product.py"""
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
    pair = graph.pair_fwd
    d, nb, n = layer.out_channels, layer.num_bases, graph.scale.numel()
    cells, xb_nb, zeros = graph.pair_buffers(n, nb, d, dev)
    run = lambda: ops.pair_product(cells, xb_nb, symmetric=pair.symmetric, links=pair.links, zeros=zeros)
    print('d=%d  pair_product %.1f us' % (d, bench.time_launch_us(run)))
    if '+debug' in _lib.build_id():
        for dbg, what in ((256, 'no cell fetches (zero block only)'), (512, 'no MFMAs'), (256 + 512, 'neither: links, XB staging, stores'),
                          (1024, 'no XB staging'), (2048, 'no stores'), (256 + 512 + 1024 + 2048, 'link words + barrier only')):
            _lib.set_option('dp_debug', dbg)
            print('   %-46s %.1f us' % (what, bench.time_launch_us(run)))
        _lib.set_option('dp_debug', 0)
