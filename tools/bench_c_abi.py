"""What a foreign host gets from the op-level C ABI at BioSNAP size: the two D-D R-GCN layers of the encoder (64 -> 32 -> 16,
32 bases, R = 1 097) forward + backward through `tipk_graph_build` / `tipk_rgcn_fwd` / `tipk_rgcn_bwd` alone, captured into a
hipGraph and replayed.  Compare with the `dd_launches_us` of bench.py (the same two layers inside the PyTorch modules).

    python tools/bench_c_abi.py [--steps 50]
"""
import argparse
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'examples'))
import c_abi_host as host                                      # noqa: E402  (ctypes signatures + the Layer wrapper)
from tip_amd.data import build_data_dict                        # noqa: E402  (the synthetic BioSNAP-shaped graph only)


def whole_encoder(lib, dd, gd, dev, steps):
    """FMEncoder.forward (mod = 'cat', src/layers.py:520-550) + its backward from a fixed upstream gradient, every pass one op-level
    C call: tipk_gcn_fwd x 2 (identity features), tipk_hier_fwd, tipk_rows_affine (embed / d_norm into the left columns of the
    concatenation), tipk_rgcn_fwd x 2 (pair form) -- and back."""
    ptr, ok = host.ptr, host.ok
    n_d, n_p, r = dd['n_drug'], dd['n_prot'], dd['n_dd_et']
    pp, dp = dd['pp_train_indices'].to(dev), dd['dp_edge_index'].to(dev)
    gp, gh = C.c_void_p(), C.c_void_p()
    ok(lib, lib.tipk_gcn_graph_build(ptr(pp), 8, pp.shape[1], n_p, C.byref(gp)), 'gcn graph')
    ok(lib, lib.tipk_hier_graph_build(ptr(dp), 8, dp.shape[1], n_p + n_d, n_p, C.byref(gh)), 'hier graph')
    torch.manual_seed(1)
    mk = lambda *s: (torch.randn(*s, device=dev) * 0.1).contiguous()
    d1, d2, dh, ne = 32, 16, 16, 48
    w1t, b1, w2, b2, wh, embed = mk(n_p, d1), mk(d1), mk(d2, d1), mk(d2), mk(d2, dh), mk(n_d, ne)
    d_norm = torch.ones(n_d, device=dev)
    l1 = host.Layer(lib, gd, mk(32, ne + dh, 32), mk(r, 32), mk(ne + dh, 32), dev, True)
    l2 = host.Layer(lib, gd, mk(32, 32, 16), mk(r, 32), mk(32, 16), dev, True)
    ws = torch.empty(max(lib.tipk_gcn_workspace_bytes(gp, n_p, d1), lib.tipk_gcn_workspace_bytes(gp, d1, d2),
                         lib.tipk_hier_workspace_bytes(gh, d2, dh)), dtype=torch.uint8, device=dev)
    e = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
    h1, h2all, x0, gz = e(n_p, d1), torch.zeros(n_p + n_d, d2, device=dev), e(n_d, ne + dh), mk(n_d, 16)
    g_embed, g_h2all, g_wh, g_h1, g_w2, g_b2, g_w1t, g_b1 = e(n_d, ne), e(n_p + n_d, d2), e(d2, dh), e(n_p, d1), e(d2, d1), e(d2), e(n_p, d1), e(d1)
    W = ne + dh

    def step(st):
        ok(lib, lib.tipk_gcn_fwd(gp, None, 0, n_p, ptr(w1t), 1, d1, ptr(b1), d1, 1, ptr(h1), d1, ptr(ws), ws.numel(), st), 'conv1')
        ok(lib, lib.tipk_gcn_fwd(gp, ptr(h1), d1, d1, ptr(w2), d1, 1, ptr(b2), d2, 0, ptr(h2all), d2, ptr(ws), ws.numel(), st), 'conv2')
        pd = x0[:, ne:]                                                     # cat(embed / d_norm, P->D) : two strided writes
        ok(lib, lib.tipk_hier_fwd(gh, ptr(h2all), d2, d2, ptr(wh), dh, ptr(pd), W, ptr(ws), ws.numel(), st), 'hier')
        ok(lib, lib.tipk_rows_affine(ptr(embed), ne, None, ptr(d_norm), None, 0, ptr(x0), W, n_d, ne, 0, st), 'embed / d_norm')
        l2.forward(l1.forward(x0, relu=True))
        g_x0 = l1.backward(l2.backward(gz)[0])[0]
        ok(lib, lib.tipk_rows_affine(ptr(g_x0), W, None, ptr(d_norm), None, 0, ptr(g_embed), ne, n_d, ne, 0, st), 'd embed')
        g_pd = g_x0[:, ne:]
        ok(lib, lib.tipk_hier_bwd(gh, ptr(h2all), d2, d2, ptr(wh), dh, ptr(g_pd), W, ptr(g_h2all), d2, ptr(g_wh), ptr(ws), ws.numel(), st), 'hier bwd')
        ok(lib, lib.tipk_gcn_bwd(gp, ptr(h1), d1, d1, ptr(w2), d1, 1, d2, ptr(g_h2all), d2, None, 0, ptr(g_h1), d1, ptr(g_w2), d1, 1, ptr(g_b2),
                                 ptr(ws), ws.numel(), st), 'conv2 bwd')
        ok(lib, lib.tipk_gcn_bwd(gp, None, 0, n_p, ptr(w1t), 1, d1, d1, ptr(g_h1), d1, ptr(h1), d1, None, 0, ptr(g_w1t), 1, d1, ptr(g_b1),
                                 ptr(ws), ws.numel(), st), 'conv1 bwd')

    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        st = C.c_void_p(s.cuda_stream)
        l1.stream = l2.stream = st
        for _ in range(3):
            step(st)
        s.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            step(st)
    for _ in range(5):
        gr.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        gr.replay()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    assert bool(torch.isfinite(g_w1t).all()) and bool(torch.isfinite(g_embed).all())
    del gr
    for h in (gp, gh):
        ok(lib, lib.tipk_graph_destroy(h), 'tipk_graph_destroy')
    return {'ms_per_step': ms, 'edges_per_s': int(dd['dd_train_idx'].shape[1]) / ms * 1e3, 'routes': [l1.route, l2.route],
            'what': 'FMEncoder (cat) forward + backward, every pass one op-level C call, hipGraph replay'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--json', action='store_true', help='one JSON object on the last line (bench.py reads it)')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    lib = host.load_library()
    dd = build_data_dict()
    ei, rg = dd['dd_train_idx'].to(dev), dd['dd_train_range'].to(dev)
    n, r, e = dd['n_drug'], dd['n_dd_et'], int(ei.shape[1])
    t0 = time.perf_counter()
    g = host.build_graph(lib, ei, None, rg, n, r)
    build_s = time.perf_counter() - t0
    rec = {'workload': 'both D-D R-GCN layers (64 -> 32 -> 16, 32 bases) fwd + bwd, N = %d, R = %d, E = %d' % (n, r, e), 'graph_build_s': build_s}
    for fast in (False, True):
        torch.manual_seed(0)
        mk = lambda *s: torch.randn(*s, device=dev) * 0.1
        t0 = time.perf_counter()
        l1 = host.Layer(lib, g, mk(32, 64, 32), mk(r, 32), mk(64, 32), dev, fast)
        l2 = host.Layer(lib, g, mk(32, 32, 16), mk(r, 32), mk(32, 16), dev, fast)
        torch.cuda.synchronize()
        prep_s = time.perf_counter() - t0
        x, gz = mk(n, 64), mk(n, 16)

        def step():
            h = l1.forward(x, relu=True)
            l2.forward(h)
            gh = l2.backward(gz)[0]
            l1.backward(gh)

        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            l1.stream = l2.stream = C.c_void_p(s.cuda_stream)
            for _ in range(3):
                step()
            s.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                step()
        for _ in range(5):
            gr.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            gr.replay()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        print('op-level C ABI, both D-D layers fwd + bwd at BioSNAP size (N = %d, R = %d, E = %d), routes %d / %d: %.3f ms per step = '
              '%.2f G edges/s; tipk_graph_build %.2f s, layers + tipk_graph_prepare_rgcn %.2f s'
              % (n, r, e, l1.route, l2.route, ms, e / ms / 1e6, build_s, prep_s))
        rec['pair_form' if fast else 'generic'] = {'routes': [l1.route, l2.route], 'ms_per_step': ms, 'edges_per_s': e / ms * 1e3,
                                                   'layers_and_prepare_s': prep_s}
        del gr, l1, l2
    rec['encoder'] = whole_encoder(lib, dd, g, dev, args.steps)
    print('op-level C ABI, WHOLE encoder fwd + bwd (P-P GCN x 2, P->D, mix, R-GCN x 2 in pair form): %.3f ms per step = %.2f G edges/s'
          % (rec['encoder']['ms_per_step'], e / rec['encoder']['ms_per_step'] / 1e6))
    host.ok(lib, lib.tipk_graph_destroy(g), 'tipk_graph_destroy')
    if args.json:
        import json
        print(json.dumps(rec))


if __name__ == '__main__':
    main()
