"""Config-5-shaped timing of tipk_rgcn_row_products (graph-timed), with the debug decomposition on a debug build
(TIPK_LIB=tip_amd/libtipk_debug.so).   python3 tools/bench_row_products.py [n_drug n_rel n_edges]"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tip_amd import ops, _lib
from tip_amd.data import synthetic_data_dict
from tip_amd.plan import build_row_stream_plan
n, r, e = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (10000, 2000, 50_000_000)
dev = 'cuda:0'
dd = synthetic_data_dict(n_drug=n, n_rel=r, n_edges=e)
src, dst = dd['dd_train_idx'].to(dev)
rel = dd['dd_train_et'].to(dev)
t0 = time.perf_counter()
rp = build_row_stream_plan(dst, src, rel, n, r)
torch.cuda.synchronize()
nb_ = rp.desc[..., 1].long()
print('plan built in %.1f s: %d batches (%.1f MB), per tile mean %.2f max %d; padding %.2f x' % (
    time.perf_counter() - t0, rp.entries.shape[0], rp.entries.numel() * 4 / 1e6, float(nb_.float().mean()), int(nb_.max()),
    rp.entries.shape[0] * 32 / max(1, rel.numel())))
ch, nb = 128, 32
x = torch.randn(n, ch, device=dev); att = torch.randn(r, nb, device=dev) * 0.1; xb = torch.randn(nb, n * ch, device=dev)
def t(f, reps=3):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
from tip_amd.plan import build_row_stream_plan_s
rs = build_row_stream_plan_s(dst, src, rel, n, r)
print('wave-uniform entries: %d batches, padding %.2f x' % (rs.entries.shape[0], rs.entries.shape[0] * 16 / max(1, rel.numel())))
print('S forward  (T)        ms %.3f' % t(lambda: ops.row_products(rs, x, att)))
print('S backward (T + datt) ms %.3f' % t(lambda: ops.row_products(rs, x, att, xb)))
ts_ = ops.row_products(rs, x, att)
print('forward  (T)        ms %.3f' % t(lambda: ops.row_products(rp, x, att)))
print('backward (T + datt) ms %.3f' % t(lambda: ops.row_products(rp, x, att, xb)))
if '+debug' in _lib.build_id():
    for bits, name in ((1, 'no table loads'), (2, 'no LDS adds'), (3, 'neither'), (4, 'no product 1'), (7, 'entry words + zeroing only'), (8, 'plain read-add-write')):
        _lib.set_option('dp_debug', bits)
        print('%-28s forward ms %.3f' % (name, t(lambda: ops.row_products(rp, x, att))))
    _lib.set_option('dp_debug', 0)
# sampled check against the definition
tt = ops.row_products(rp, x, att)
print('S vs V: max |dT| / max |T| = %.2e' % float((ts_ - tt).abs().max() / tt.abs().max()))
for v in (0, n // 2, n - 1):
    m = dst == v
    s = torch.zeros(r, ch, dtype=torch.float64, device=dev)
    s.index_add_(0, rel[m], x[src[m]].double())
    want = att.double().t() @ s
    print('node %d: max |T - def| / max |def| = %.2e' % (v, float((tt[:, v].double() - want).abs().max() / want.abs().max())))
