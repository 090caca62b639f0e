"""Where the wave-stream gather spends its cycles (debug library: make -C tip_amd/csrc debug; TIPK_LIB=tip_amd/libtipk_debug.so):
per wave { lifetime, table staging + barrier, band loop, bands } for the step's own D-D launches."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from tip_amd import _lib
from tip_amd.data import build_data_dict
from tip_amd.layers import TIP, Setting
dev = torch.device('cuda:0')
model = TIP(Setting(), dev, data=build_data_dict())
enc, d = model.encoder, model.data
enc(d.d_feat, d.dd_train_idx, d.dd_train_et, d.dd_train_range, d.d_norm, d.p_feat, d.pp_train_indices, d.dp_edge_index, d.dp_range_list)
L = _lib.lib()
buf = (C.c_ulonglong * (4096 * 4))()
for rec in bench.dd_launches(enc, dev):
    if not rec.get('aggregation'):
        continue
    label = rec['label']
    for _ in range(3):
        rec['fn']()
    torch.cuda.synchronize()
    assert L.tipk_debug_rs_stamps(buf) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 4).astype(np.float64)
    tot, stage, loop, bands = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
    print(label)
    print('   wave lifetime   mean %7.0f  min %7.0f  max %7.0f' % (tot.mean(), tot.min(), tot.max()))
    print('   table staging   mean %7.0f  (%.0f %%)' % (stage.mean(), 100 * stage.mean() / tot.mean()))
    print('   band loop       mean %7.0f  min %7.0f  max %7.0f   (max wave / mean %.2f)' % (loop.mean(), loop.min(), loop.max(), loop.max() / loop.mean()))
    print('   bands per wave  mean %5.1f  max %3.0f;  cycles per band %.0f' % (bands.mean(), bands.max(), loop.sum() / max(bands.sum(), 1)))
    w = a.reshape(256, 16, 4)
    print('   per-workgroup lifetime (slowest wave): mean %7.0f  max %7.0f' % (w[:, :, 0].max(1).mean(), w[:, :, 0].max(1).max()))
