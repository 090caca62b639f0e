"""The typed negative sampler at BioSNAP size, graph-timed; with the debug build (TIPK_LIB=tip_amd/libtipk_debug.so) its parts:
   python tools/bench_sampler.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tip_amd import _lib
from tip_amd.data import build_data_dict
from tip_amd.neg_sampling import typed_negative_sampling
dd = build_data_dict()
dev = 'cuda:0'
pos, rg = dd['dd_train_idx'].to(dev), dd['dd_train_range'].to(dev)
f = lambda: typed_negative_sampling(pos, 645, rg, packed=True)
print('sampler (packed pairs, %d positions) %.1f us' % (pos.shape[1], bench.time_launch_us(f)))
if '+debug' in _lib.build_id():
    for bits, what in ((1, 'bitmaps only (no draws)'), (2, 'draws only (no bitmaps)'), (3, 'neither: unit walk + stream advance')):
        _lib.set_option('dm_debug', bits)
        print('   %-40s %.1f us' % (what, bench.time_launch_us(f)))
    _lib.set_option('dm_debug', 0)
