#!/usr/bin/env python3
"""A host that is NOT the tip_amd package: one R-GCN layer forward + backward through the op-level C ABI alone.

    python examples/c_abi_host.py [fixture.npz ...]        (default: tests/golden/rgcn_sym.npz rgcn_directed.npz)

What a maintainer of the reference would write to route `MyRGCNConv2.forward` (src/layers.py:157-188) and its autograd to
libtipk.so from any language with a C FFI: `tipk_graph_build` once per graph, `tipk_rgcn_fwd` / `tipk_rgcn_bwd` per pass,
`tipk_graph_destroy` at the end (include/tipk.h section 10) -- and the same for the other two layer kinds of the path, `GCNConv`
as `PPEncoder` uses it (`tipk_gcn_*`, identity and dense features) and `MyHierarchyConv` (`tipk_hier_*`).  Only ctypes + torch-for-device-memory are used here: neither
`tip_amd.ops` nor `tip_amd.plan` (nor any other module of the package) is imported -- asserted at the end.  The fixtures are
outputs and autograd gradients of the reference's own layers (oracle/make_golden.py); both the range-list form (MyRGCNConv2)
and the edge-type form (MyRGCNConv) of the graph are built.
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P, I, L = C.c_void_p, C.c_int, C.c_int64
ROUTES = set()                # routes the R-GCN layers of the current fixture took (0 generic | 1 | 2 pair form)


def load_library():
    lib = C.CDLL(os.path.join(ROOT, 'tip_amd', 'libtipk.so'))
    lib.tipk_strerror.restype = C.c_char_p
    lib.tipk_strerror.argtypes = [I]
    lib.tipk_graph_build.restype = I
    lib.tipk_graph_build.argtypes = [P, P, P, I, L, L, L, P, C.POINTER(P)]
    lib.tipk_graph_destroy.restype = I
    lib.tipk_graph_destroy.argtypes = [P]
    lib.tipk_rgcn_workspace_bytes.restype = L
    lib.tipk_rgcn_workspace_bytes.argtypes = [P, I, I, I]
    lib.tipk_rgcn_fwd.restype = I
    lib.tipk_rgcn_fwd.argtypes = [P, P, L, I, P, P, P, I, I, I, P, L, P, L, P]
    lib.tipk_rgcn_bwd.restype = I
    lib.tipk_rgcn_bwd.argtypes = [P, P, L, I, P, P, P, I, I, P, L, P, L, P, L, P, P, P, P, L, P]
    lib.tipk_rgcn_bwd_ex.restype = I
    lib.tipk_rgcn_bwd_ex.argtypes = [P, P, L, I, P, P, P, I, I, P, L, P, L, P, L, P, P, P, P, L, I, P]
    lib.tipk_graph_prepare_rgcn.restype = I
    lib.tipk_graph_prepare_rgcn.argtypes = [P, I, I]
    lib.tipk_graph_rgcn_route.restype = I
    lib.tipk_graph_rgcn_route.argtypes = [P, I, I]
    lib.tipk_gcn_graph_build.restype = I
    lib.tipk_gcn_graph_build.argtypes = [P, I, L, L, C.POINTER(P)]
    lib.tipk_gcn_workspace_bytes.restype = L
    lib.tipk_gcn_workspace_bytes.argtypes = [P, I, I]
    lib.tipk_gcn_fwd.restype = I
    lib.tipk_gcn_fwd.argtypes = [P, P, L, I, P, L, L, P, I, I, P, L, P, L, P]
    lib.tipk_gcn_bwd.restype = I
    lib.tipk_gcn_bwd.argtypes = [P, P, L, I, P, L, L, I, P, L, P, L, P, L, P, L, L, P, P, L, P]
    lib.tipk_hier_graph_build.restype = I
    lib.tipk_hier_graph_build.argtypes = [P, I, L, L, L, C.POINTER(P)]
    lib.tipk_hier_workspace_bytes.restype = L
    lib.tipk_hier_workspace_bytes.argtypes = [P, I, I]
    lib.tipk_hier_fwd.restype = I
    lib.tipk_hier_fwd.argtypes = [P, P, L, I, P, I, P, L, P, L, P]
    lib.tipk_hier_bwd.restype = I
    lib.tipk_hier_bwd.argtypes = [P, P, L, I, P, I, P, L, P, L, P, P, L, P]
    lib.tipk_rows_affine.restype = I
    lib.tipk_rows_affine.argtypes = [P, L, P, P, P, L, P, L, L, L, I, P]
    lib.tipk_distmult_loss.restype = I
    lib.tipk_distmult_loss.argtypes = [P, L, I, P, L, P, P, P, P, I, P, I, L, P, L, P, P, P, P, P]
    return lib


def ok(lib, status, what):
    if status != 0:
        raise RuntimeError('%s: %s (%d)' % (what, lib.tipk_strerror(status).decode(), status))


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class Layer(object):
    """One R-GCN layer on a graph handle: forward(x, relu) / backward(grad_out).
    fast: ask the handle for the LDS-resident pair form of this layer shape (`tipk_graph_prepare_rgcn`; the generic route stays
    where the graph or the shape does not qualify); `route` = 0 generic | 1 pair-form forward | 2 pair form both ways."""

    def __init__(self, lib, graph, basis, att, root, dev, fast=False):
        self.lib, self.graph = lib, graph
        self.basis, self.att, self.root = (t.to(dev).contiguous() for t in (basis, att, root))
        self.nb, self.d_in, self.d_out = basis.shape
        if fast:
            st = lib.tipk_graph_prepare_rgcn(graph, self.nb, self.d_out)
            if st != -2:                                                  # TIPK_EUNSUPPORTED: not a pair-form shape
                ok(lib, st, 'tipk_graph_prepare_rgcn')
        self.route = lib.tipk_graph_rgcn_route(graph, self.nb, self.d_out)
        n = lib.tipk_rgcn_workspace_bytes(graph, self.d_in, self.d_out, self.nb)      # (after prepare: the route sizes it)
        assert n > 0
        self.ws = torch.empty(n, dtype=torch.uint8, device=dev)           # caller-owned, reused by both passes
        self.stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        self.ws_from_fwd = False

    def forward(self, x, relu=False):
        self.x, self.relu = x, relu
        self.out = torch.empty((x.shape[0], self.d_out), dtype=torch.float32, device=x.device)
        ok(self.lib, self.lib.tipk_rgcn_fwd(self.graph, ptr(x), x.stride(0), self.d_in, ptr(self.basis), ptr(self.att), ptr(self.root),
                                            self.nb, self.d_out, int(relu), ptr(self.out), self.out.stride(0), ptr(self.ws),
                                            self.ws.numel(), self.stream), 'tipk_rgcn_fwd')
        self.ws_from_fwd = True                                           # nothing else runs on this layer's workspace
        return self.out

    def backward(self, g):
        x = self.x
        gx = torch.empty_like(x)
        gb, ga, gr = torch.empty_like(self.basis), torch.empty_like(self.att), torch.empty_like(self.root)
        gate = self.out if self.relu else None
        flags = 1 if self.ws_from_fwd else 0                              # TIPK_RGCN_WORKSPACE_FROM_FWD
        ok(self.lib, self.lib.tipk_rgcn_bwd_ex(self.graph, ptr(x), x.stride(0), self.d_in, ptr(self.basis), ptr(self.att), ptr(self.root),
                                               self.nb, self.d_out, ptr(g), g.stride(0), ptr(gate), self.d_out if self.relu else 0,
                                               ptr(gx), gx.stride(0), ptr(gb), ptr(ga), ptr(gr), ptr(self.ws), self.ws.numel(), flags,
                                               self.stream), 'tipk_rgcn_bwd_ex')
        self.ws_from_fwd = False
        return gx, gb, ga, gr


def build_graph(lib, ei, et, rg, n, r):
    h = C.c_void_p()
    ok(lib, lib.tipk_graph_build(ptr(ei), ptr(et), ptr(rg), 8, ei.shape[1], n, r, None, C.byref(h)), 'tipk_graph_build')
    return h


def check(name, got, want, rtol, scale):
    got, want = got.detach().cpu().double(), torch.as_tensor(want).double()
    err = float((got - want).abs().max())
    lim = rtol * float(want.abs().max()) * scale + 1e-12
    if not (err <= lim and bool(torch.isfinite(got).all())):
        raise SystemExit('%s: max |diff| %.3e > %.3e' % (name, err, lim))
    return err / max(1e-30, float(want.abs().max()))


def run_single(lib, path, dev, fast=False):
    g = {k: torch.from_numpy(v) if v.dtype.kind in 'fi' and v.ndim else v for k, v in np.load(path).items()}
    x = g['x'].to(dev).contiguous()
    n, r = x.shape[0], g['att'].shape[0]
    ei, et, rg = g['dd_idx'].to(dev).contiguous(), g['dd_et'].to(dev).contiguous(), g['dd_range'].to(dev).contiguous()
    worst = 0.0
    for form in ('range_list', 'edge_type'):                                # MyRGCNConv2 | MyRGCNConv
        graph = build_graph(lib, ei, None if form == 'range_list' else et, rg if form == 'range_list' else None, n, r)
        layer = Layer(lib, graph, g['basis'], g['att'], g['root'], dev, fast)
        ROUTES.add(layer.route)
        out = layer.forward(x)
        gx, gb, ga, gr = layer.backward(g['upstream'].to(dev).contiguous())
        torch.cuda.synchronize()
        for name, got, want in (('out', out, g['out']), ('grad_x', gx, g['grad_x']), ('grad.basis', gb, g['grad.basis']),
                                ('grad.att', ga, g['grad.att']), ('grad.root', gr, g['grad.root'])):
            worst = max(worst, check('%s[%s] %s' % (os.path.basename(path), form, name), got, want, 2e-5, 1.0))
        out2 = layer.forward(x)
        assert torch.equal(out2, out), 'not bitwise reproducible'
        again = layer.backward(g['upstream'].to(dev).contiguous())          # workspace from THIS forward pass
        layer.ws.fill_(255)                                                   # ... and from nothing: recomputed (NaN bit patterns)
        cold = layer.backward(g['upstream'].to(dev).contiguous())
        for a, b, c in zip((gx, gb, ga, gr), again, cold):
            assert torch.equal(a, b) and torch.equal(a, c), 'backward pass depends on the workspace'
        ok(lib, lib.tipk_graph_destroy(graph), 'tipk_graph_destroy')
    return worst


def run_two_layers(lib, path, dev, fast=False):
    """64 -> 32 (ReLU inside the layer's last kernel) -> 16 on the nasty 61-drug graph, gradients through both layers."""
    g = {k: torch.from_numpy(v) if v.dtype.kind in 'fi' and v.ndim else v for k, v in np.load(path).items()}
    x = g['x'].to(dev).contiguous()
    n, r = x.shape[0], g['l1.att'].shape[0]
    graph = build_graph(lib, g['dd_idx'].to(dev).contiguous(), None, g['dd_range'].to(dev).contiguous(), n, r)
    l1 = Layer(lib, graph, g['l1.basis'], g['l1.att'], g['l1.root'], dev, fast)
    l2 = Layer(lib, graph, g['l2.basis'], g['l2.att'], g['l2.root'], dev, fast)
    ROUTES.update((l1.route, l2.route))
    if fast:                                                                  # reference dims (32 bases, 64 -> 32 -> 16): pair form both ways
        assert (l1.route, l2.route) == (2, 2), (l1.route, l2.route)
    out = l2.forward(l1.forward(x, relu=True))
    gx1, gb2, ga2, gr2 = l2.backward(g['upstream'].to(dev).contiguous())
    gx, gb1, ga1, gr1 = l1.backward(gx1)
    torch.cuda.synchronize()
    worst = 0.0
    for name, got, want in (('out', out, g['out']), ('grad_x', gx, g['grad_x']), ('l1.basis', gb1, g['grad.l1.basis']),
                            ('l1.att', ga1, g['grad.l1.att']), ('l1.root', gr1, g['grad.l1.root']), ('l2.basis', gb2, g['grad.l2.basis']),
                            ('l2.att', ga2, g['grad.l2.att']), ('l2.root', gr2, g['grad.l2.root'])):
        worst = max(worst, check('%s %s' % (os.path.basename(path), name), got, want, 2e-5, 1.0))
    ok(lib, lib.tipk_graph_destroy(graph), 'tipk_graph_destroy')
    return worst


def load(path):
    return {k: torch.from_numpy(v) if v.dtype.kind in 'fi' and v.ndim else v for k, v in np.load(path).items()}


def run_pp_encoder(lib, path, dev):
    """PPEncoder = GCNConv(n_prot, 32) + ReLU + GCNConv(32, 16) on identity features (src/layers.py:380-395) through
    tipk_gcn_graph_build / tipk_gcn_fwd / tipk_gcn_bwd; the dense-feature fixture runs the same two layers on x [n, 24]."""
    g = load(path)
    dense = 'x' in g
    n = int(g['x'].shape[0]) if dense else int(g['n_prot'])
    ei = g['pp_idx'].to(dev).contiguous()
    h = C.c_void_p()
    ok(lib, lib.tipk_gcn_graph_build(ptr(ei), 8, ei.shape[1], n, C.byref(h)), 'tipk_gcn_graph_build')
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    w1, b1, w2, b2 = (g[k].to(dev) for k in ('conv1.lin.weight', 'conv1.bias', 'conv2.lin.weight', 'conv2.bias'))
    d1, d2 = w1.shape[0], w2.shape[0]
    ws = torch.empty(max(lib.tipk_gcn_workspace_bytes(h, n, d1), lib.tipk_gcn_workspace_bytes(h, d1, d2)), dtype=torch.uint8, device=dev)
    h1 = torch.empty(n, d1, device=dev)
    out = torch.empty(n, d2, device=dev)
    up = g['upstream'].to(dev).contiguous()
    g_h1, g_w2, g_b2, g_b1 = torch.empty(n, d1, device=dev), torch.empty(d2, d1, device=dev), torch.empty(d2, device=dev), torch.empty(d1, device=dev)
    if dense:
        x = g['x'].to(dev).contiguous()
        w1c = w1.contiguous()                                                 # [out, in] memory: w_so = in, w_si = 1
        g_w1 = torch.empty_like(w1c)
        g_x = torch.empty_like(x)
        ok(lib, lib.tipk_gcn_fwd(h, ptr(x), x.stride(0), x.shape[1], ptr(w1c), w1c.stride(0), 1, ptr(b1), d1, 1, ptr(h1), d1, ptr(ws), ws.numel(), st), 'gcn_fwd 1')
    else:
        w1t = w1.t().contiguous()                                             # [in, out] memory: lin(I) = W^T is read in place
        g_w1t = torch.empty_like(w1t)
        ok(lib, lib.tipk_gcn_fwd(h, None, 0, n, ptr(w1t), 1, d1, ptr(b1), d1, 1, ptr(h1), d1, ptr(ws), ws.numel(), st), 'gcn_fwd 1')
    w2c = w2.contiguous()
    ok(lib, lib.tipk_gcn_fwd(h, ptr(h1), d1, d1, ptr(w2c), d1, 1, ptr(b2), d2, 0, ptr(out), d2, ptr(ws), ws.numel(), st), 'gcn_fwd 2')
    ok(lib, lib.tipk_gcn_bwd(h, ptr(h1), d1, d1, ptr(w2c), d1, 1, d2, ptr(up), d2, None, 0, ptr(g_h1), d1, ptr(g_w2), d1, 1, ptr(g_b2),
                             ptr(ws), ws.numel(), st), 'gcn_bwd 2')
    if dense:
        ok(lib, lib.tipk_gcn_bwd(h, ptr(x), x.stride(0), x.shape[1], ptr(w1c), w1c.stride(0), 1, d1, ptr(g_h1), d1, ptr(h1), d1, ptr(g_x),
                                 g_x.stride(0), ptr(g_w1), g_w1.stride(0), 1, ptr(g_b1), ptr(ws), ws.numel(), st), 'gcn_bwd 1')
    else:
        ok(lib, lib.tipk_gcn_bwd(h, None, 0, n, ptr(w1t), 1, d1, d1, ptr(g_h1), d1, ptr(h1), d1, None, 0, ptr(g_w1t), 1, d1, ptr(g_b1),
                                 ptr(ws), ws.numel(), st), 'gcn_bwd 1')
        g_w1 = g_w1t.t()
    torch.cuda.synchronize()
    worst = 0.0
    checks = [('out', out, g['out']), ('conv1.lin.weight', g_w1, g['grad.conv1.lin.weight'])]
    if not dense:
        checks += [('conv2.lin.weight', g_w2, g['grad.conv2.lin.weight']), ('conv2.bias', g_b2, g['grad.conv2.bias']),
                   ('conv1.bias', g_b1, g['grad.conv1.bias'])]
    else:
        checks += [('grad_x', g_x, g['grad_x'])]
    for name, got, want in checks:
        worst = max(worst, check('%s %s' % (os.path.basename(path), name), got, want, 2e-5, 1.0))
    ok(lib, lib.tipk_graph_destroy(h), 'tipk_graph_destroy')
    return worst


def run_hier(lib, path, dev):
    """MyHierarchyConv (src/layers.py:196-247) through tipk_hier_graph_build / tipk_hier_fwd / tipk_hier_bwd."""
    g = load(path)
    x = g['x'].to(dev).contiguous()
    n_all, d_in = x.shape
    n_src = int(g['n_source'])
    w = g['weight'].to(dev).contiguous()
    d_out = w.shape[1]
    ei = g['dp_idx'].to(dev).contiguous()
    h = C.c_void_p()
    ok(lib, lib.tipk_hier_graph_build(ptr(ei), 8, ei.shape[1], n_all, n_src, C.byref(h)), 'tipk_hier_graph_build')
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ws = torch.empty(lib.tipk_hier_workspace_bytes(h, d_in, d_out), dtype=torch.uint8, device=dev)
    out = torch.empty(n_all - n_src, d_out, device=dev)
    up = g['upstream'].to(dev).contiguous()
    g_x, g_w = torch.empty_like(x), torch.empty_like(w)
    ok(lib, lib.tipk_hier_fwd(h, ptr(x), d_in, d_in, ptr(w), d_out, ptr(out), d_out, ptr(ws), ws.numel(), st), 'tipk_hier_fwd')
    ok(lib, lib.tipk_hier_bwd(h, ptr(x), d_in, d_in, ptr(w), d_out, ptr(up), d_out, ptr(g_x), d_in, ptr(g_w), ptr(ws), ws.numel(), st),
       'tipk_hier_bwd')
    torch.cuda.synchronize()
    worst = 0.0
    for name, got, want in (('out', out, g['out']), ('grad_x', g_x, g['grad_x']), ('grad.weight', g_w, g['grad.weight'])):
        worst = max(worst, check('%s %s' % (os.path.basename(path), name), got, want, 2e-5, 1.0))
    ok(lib, lib.tipk_graph_destroy(h), 'tipk_graph_destroy')
    return worst


def run_tip_step(lib, path, dev, fast=False):
    """The WHOLE training step of TIP (mod = 'add'; src/layers.py:328-342 on top of :520-550) through the op-level C ABI alone:
    P-P GCN x 2 -> P -> D mean . W -> embed / d_norm + . -> R-GCN (ReLU) -> R-GCN -> DistMult objective, and every gradient
    back, against the reference's own loss and autograd gradients on the negatives its sampler drew (`tip_add_small`)."""
    g = load(path)
    f = lambda k: g[k].to(dev).contiguous()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    n_d, n_p, R = int(g['n_drug']), int(g['n_prot']), int(g['n_dd_et'])
    E = g['dd_train_idx'].shape[1]
    zeros = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)
    empty = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    handle = lambda: C.c_void_p()
    # graphs: once
    gp, gh, gd = handle(), handle(), handle()
    pp, dp, dd, rg = f('pp_train_indices'), f('dp_edge_index'), f('dd_train_idx'), f('dd_train_range')
    ok(lib, lib.tipk_gcn_graph_build(ptr(pp), 8, pp.shape[1], n_p, C.byref(gp)), 'gcn graph')
    ok(lib, lib.tipk_hier_graph_build(ptr(dp), 8, dp.shape[1], n_p + n_d, n_p, C.byref(gh)), 'hier graph')
    ok(lib, lib.tipk_graph_build(ptr(dd), None, ptr(rg), 8, E, n_d, R, None, C.byref(gd)), 'dd graph')
    w1t = g['encoder.pp_encoder.conv1.lin.weight'].t().contiguous().to(dev)       # [in, out] memory: lin(I) = W^T in place
    b1, w2, b2 = f('encoder.pp_encoder.conv1.bias'), f('encoder.pp_encoder.conv2.lin.weight'), f('encoder.pp_encoder.conv2.bias')
    wh, embed, d_norm = f('encoder.hgcn.weight'), f('encoder.embed'), f('d_norm')
    l1 = Layer(lib, gd, g['encoder.rgcn1.basis'], g['encoder.rgcn1.att'], g['encoder.rgcn1.root'], dev, fast)
    l2 = Layer(lib, gd, g['encoder.rgcn2.basis'], g['encoder.rgcn2.att'], g['encoder.rgcn2.root'], dev, fast)
    ROUTES.update((l1.route, l2.route))
    wdec = f('decoder.weight')
    d1, d2, dh = w1t.shape[1], w2.shape[0], wh.shape[1]
    ws = torch.empty(max(lib.tipk_gcn_workspace_bytes(gp, n_p, d1), lib.tipk_gcn_workspace_bytes(gp, d1, d2),
                         lib.tipk_hier_workspace_bytes(gh, d2, dh)), dtype=torch.uint8, device=dev)
    # ---- forward
    h1, h2all = empty(n_p, d1), zeros(n_p + n_d, d2)                          # (the drug rows of the concatenation stay zero: :526)
    ok(lib, lib.tipk_gcn_fwd(gp, None, 0, n_p, ptr(w1t), 1, d1, ptr(b1), d1, 1, ptr(h1), d1, ptr(ws), ws.numel(), st), 'conv1')
    ok(lib, lib.tipk_gcn_fwd(gp, ptr(h1), d1, d1, ptr(w2), d1, 1, ptr(b2), d2, 0, ptr(h2all), d2, ptr(ws), ws.numel(), st), 'conv2')
    pd = empty(n_d, dh)
    ok(lib, lib.tipk_hier_fwd(gh, ptr(h2all), d2, d2, ptr(wh), dh, ptr(pd), dh, ptr(ws), ws.numel(), st), 'hier')
    x0 = empty(n_d, dh)
    ok(lib, lib.tipk_rows_affine(ptr(embed), dh, None, ptr(d_norm), None, 0, ptr(x0), dh, n_d, dh, 0, st), 'embed / d_norm')
    ok(lib, lib.tipk_rows_affine(ptr(pd), dh, None, None, None, 0, ptr(x0), dh, n_d, dh, 1, st), '+ pd')
    x1 = l1.forward(x0, relu=True)
    z = l2.forward(x1)
    loss, g_z, g_wdec = zeros(1), zeros(*z.shape), zeros(*wdec.shape)
    neg, et = f('train_neg'), f('dd_train_et')
    ok(lib, lib.tipk_distmult_loss(ptr(z), n_d, z.shape[1], ptr(wdec), R, ptr(dd[0]), ptr(dd[1]), ptr(neg[0]), ptr(neg[1]), 8, ptr(et), 8, E,
                                   None, 0, ptr(loss), ptr(g_z), ptr(g_wdec), None, st), 'objective')
    # ---- backward
    g_x1, gb2_, ga2_, gr2_ = l2.backward(g_z)
    g_x0, gb1_, ga1_, gr1_ = l1.backward(g_x1)
    g_embed = empty(n_d, dh)
    ok(lib, lib.tipk_rows_affine(ptr(g_x0), dh, None, ptr(d_norm), None, 0, ptr(g_embed), dh, n_d, dh, 0, st), 'd embed')
    g_h2all, g_wh = empty(n_p + n_d, d2), empty(*wh.shape)
    ok(lib, lib.tipk_hier_bwd(gh, ptr(h2all), d2, d2, ptr(wh), dh, ptr(g_x0), dh, ptr(g_h2all), d2, ptr(g_wh), ptr(ws), ws.numel(), st), 'hier bwd')
    g_h1, g_w2, g_b2 = empty(n_p, d1), empty(d2, d1), empty(d2)
    ok(lib, lib.tipk_gcn_bwd(gp, ptr(h1), d1, d1, ptr(w2), d1, 1, d2, ptr(g_h2all), d2, None, 0, ptr(g_h1), d1, ptr(g_w2), d1, 1, ptr(g_b2),
                             ptr(ws), ws.numel(), st), 'conv2 bwd')
    g_w1t, g_b1 = torch.empty_like(w1t), empty(d1)
    ok(lib, lib.tipk_gcn_bwd(gp, None, 0, n_p, ptr(w1t), 1, d1, d1, ptr(g_h1), d1, ptr(h1), d1, None, 0, ptr(g_w1t), 1, d1, ptr(g_b1),
                             ptr(ws), ws.numel(), st), 'conv1 bwd')
    torch.cuda.synchronize()
    worst = check('tip step loss', loss, g['loss'].reshape(1), 2e-5, 1.0)
    worst = max(worst, check('tip step embeddings', z, g['embeddings'], 2e-5, 1.0))
    got = {'encoder.embed': g_embed, 'encoder.pp_encoder.conv1.lin.weight': g_w1t.t(), 'encoder.pp_encoder.conv1.bias': g_b1,
           'encoder.pp_encoder.conv2.lin.weight': g_w2, 'encoder.pp_encoder.conv2.bias': g_b2, 'encoder.hgcn.weight': g_wh,
           'encoder.rgcn1.basis': gb1_, 'encoder.rgcn1.att': ga1_, 'encoder.rgcn1.root': gr1_, 'encoder.rgcn2.basis': gb2_,
           'encoder.rgcn2.att': ga2_, 'encoder.rgcn2.root': gr2_, 'decoder.weight': g_wdec}
    for k, v in got.items():
        worst = max(worst, check('tip step grad.' + k, v, g['grad.' + k], 1e-4, 1.0))
    for h in (gp, gh, gd):
        ok(lib, lib.tipk_graph_destroy(h), 'tipk_graph_destroy')
    return worst


def main():
    dev = torch.device('cuda:0')
    lib = load_library()
    golden = os.path.join(ROOT, 'tests', 'golden')
    paths = sys.argv[1:] or [os.path.join(golden, f) for f in ('rgcn_sym.npz', 'rgcn_directed.npz', 'rgcn_fast_sym.npz', 'rgcn_fast_directed.npz',
                                                                 'pp_encoder.npz', 'pp_encoder_dense.npz', 'hier_conv.npz', 'tip_add_small.npz')]
    for path in paths:
        base = os.path.basename(path)
        fn = run_pp_encoder if base.startswith('pp_encoder') else run_hier if base.startswith('hier') else \
            run_tip_step if base.startswith('tip_') else run_two_layers if 'fast' in base else run_single
        if fn in (run_single, run_two_layers, run_tip_step):                  # R-GCN layers: the generic route, then the pair form
            ROUTES.clear()
            worst = max(fn(lib, path, dev, fast=False), fn(lib, path, dev, fast=True))
            print('%-26s max error %.2e of max|want|   (R-GCN routes taken: %s)' % (base, worst, sorted(ROUTES)))
        else:
            print('%-26s max error %.2e of max|want|' % (base, fn(lib, path, dev)))
    # malformed input: an id out of range is refused, as the reference raises IndexError
    bad = torch.tensor([[0, 1], [1, 99]], device=dev)
    h = C.c_void_p()
    st = lib.tipk_graph_build(ptr(bad), ptr(torch.zeros(2, dtype=torch.int64, device=dev)), None, 8, 2, 5, 1, None, C.byref(h))
    assert st != 0 and not h.value, 'out-of-range node id accepted'
    assert not any(m == 'tip_amd' or m.startswith('tip_amd.') for m in sys.modules), 'the host imported the package'
    print('C-ABI host ok')


if __name__ == '__main__':
    main()
