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
    raw = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 4)
    live = raw[:, 0] > 0
    a = a[live]
    tot, stage, loop, bands = a[:, 0], a[:, 1], a[:, 2], (raw[live, 3] & np.uint64(0xffffffff)).astype(np.float64)
    print(label)
    print('   wave lifetime   mean %7.0f  min %7.0f  max %7.0f' % (tot.mean(), tot.min(), tot.max()))
    print('   table staging   mean %7.0f  (%.0f %%)' % (stage.mean(), 100 * stage.mean() / tot.mean()))
    print('   band loop       mean %7.0f  min %7.0f  max %7.0f   (max wave / mean %.2f)' % (loop.mean(), loop.min(), loop.max(), loop.max() / loop.mean()))
    print('   bands per wave  mean %5.1f  max %3.0f;  cycles per band %.0f' % (bands.mean(), bands.max(), loop.sum() / max(bands.sum(), 1)))
    w = a.reshape(-1, 16, 4)
    wl = w[:, :, 2]
    print('   per-workgroup band loop: mean of means %7.0f, mean of slowest waves %7.0f, max %7.0f; spread inside a workgroup (max / mean) %.2f'
          % (wl.mean(), wl.max(1).mean(), wl.max(), (wl.max(1) / wl.mean(1)).mean()))
    wb = bands.reshape(-1, 16)
    print('   bands per workgroup: mean %.1f  min %.0f  max %.0f;   workgroup band-loop mean vs its bands: corr %.2f'
          % (wb.sum(1).mean(), wb.sum(1).min(), wb.sum(1).max(), np.corrcoef(wb.sum(1), wl.mean(1))[0, 1]))
    q = np.quantile(loop, [0.1, 0.5, 0.9, 0.99])
    print('   band loop quantiles 10/50/90/99 %%: %s' % ' '.join('%.0f' % v for v in q))
    by_wave = wl.mean(0)
    print('   band loop by wave index in the workgroup: ' + ' '.join('%.0f' % v for v in by_wave))
    print('   bands by wave index:                      ' + ' '.join('%.1f' % v for v in wb.mean(0)))
