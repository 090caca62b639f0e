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
    host.ok(lib, lib.tipk_graph_destroy(g), 'tipk_graph_destroy')
    if args.json:
        import json
        print(json.dumps(rec))


if __name__ == '__main__':
    main()
