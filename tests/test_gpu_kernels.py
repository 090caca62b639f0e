"""-m gpu: every libtipk kernel through the C ABI against the oracle (fp64 CPU restatement).

Tolerances: fp32 results with a different summation order than the oracle; the bar is
|got - want| <= 2e-5 * scale + 1e-4 * |want| unless a test states otherwise (north_star states no
tighter tolerance than AUROC +-0.002; see DESIGN.md "Parity").
"""
import numpy as np
import pytest
import torch

from oracle import tip_oracle as O
from oracle.philox_sampler import typed_negative_sampling_spec

pytestmark = pytest.mark.gpu

DEV = 'cuda:0'


def close(got, want, rtol=1e-4, atol=None):
    want = want.to(torch.float64)
    got = got.detach().to('cpu', torch.float64)
    if atol is None:
        atol = 2e-5 * max(1.0, float(want.abs().max()))
    torch.testing.assert_close(got, want, rtol=rtol, atol=atol)


@pytest.fixture(scope='module')
def ops():
    from tip_amd import ops as o
    return o


# ------------------------------------------------------------------ gather_sum
@pytest.mark.parametrize('d', [4, 16, 32, 64, 128, 256, 3, 6, 50])
@pytest.mark.parametrize('chunk,weighted', [(128, False), (8, True), (1, False)])
def test_gather_sum(ops, d, chunk, weighted):
    from tip_amd.plan import build_gather_plan
    g = torch.Generator().manual_seed(d * 7 + chunk)
    n_out, n_tab, E = 301, 157, 6000
    out_row = torch.randint(0, n_out - 5, (E,), generator=g)
    out_row[:1500] = 17                                                # heavy row -> split
    tab_row = torch.randint(0, n_tab, (E,), generator=g)
    w = torch.rand(E, generator=g) if weighted else None
    table = torch.randn(n_tab, d, generator=g)
    scale = torch.rand(n_out, generator=g) + 0.5
    bias = torch.randn(d, generator=g)
    plan = build_gather_plan(out_row, tab_row, n_out, n_tab, w, chunk).to(DEV)
    want = O.gather_sum(table.double(), tab_row, out_row, n_out, None if w is None else w.double())
    close(ops.gather_sum(plan, table.to(DEV)), want)
    got = ops.gather_sum(plan, table.to(DEV), row_scale=scale.to(DEV), bias=bias.to(DEV), relu=True)
    close(got, torch.relu(want * scale.double().unsqueeze(1) + bias.double()))
    # empty rows are written as zeros (+ epilogue), output buffer pre-filled with garbage
    out = torch.full((n_out, d), 7.0, device=DEV)
    ops.gather_sum(plan, table.to(DEV), out=out)
    assert float(out[-5:].abs().max()) == 0.0


@pytest.mark.parametrize('d', [4, 16, 32, 64, 128, 6])
@pytest.mark.parametrize('chunk,weighted', [(16, True), (2, False)])
def test_gather_sum_grouped_plan(ops, d, chunk, weighted):
    """group_slots plans: split rows are combined inside the workgroup (no partial buffer, no
    finalize launch) -- oracle parity, bit-reproducible, and wrong widths are refused."""
    from tip_amd.plan import build_gather_plan, group_slots_for
    g = torch.Generator().manual_seed(d * 11 + chunk)
    n_out, n_tab, E = 301, 157, 9000
    out_row = torch.randint(0, n_out - 5, (E,), generator=g)
    out_row[:2500] = 17                                                # hub row
    tab_row = torch.randint(0, n_tab, (E,), generator=g)
    w = torch.rand(E, generator=g) if weighted else None
    table = torch.randn(n_tab, d, generator=g)
    scale = torch.rand(n_out, generator=g) + 0.5
    bias = torch.randn(d, generator=g)
    G = group_slots_for(d)
    plan = build_gather_plan(out_row, tab_row, n_out, n_tab, w, chunk, group_slots=G).to(DEV)
    assert plan.n_slots == 0
    want = O.gather_sum(table.double(), tab_row, out_row, n_out, None if w is None else w.double())
    got = ops.gather_sum(plan, table.to(DEV), row_scale=scale.to(DEV), bias=bias.to(DEV), relu=True)
    close(got, torch.relu(want * scale.double().unsqueeze(1) + bias.double()))
    assert torch.equal(got, ops.gather_sum(plan, table.to(DEV), row_scale=scale.to(DEV), bias=bias.to(DEV), relu=True))
    out = torch.full((n_out, d), 7.0, device=DEV)
    ops.gather_sum(plan, table.to(DEV), out=out)
    close(out, want)
    assert float(out[-5:].abs().max()) == 0.0
    if d == 128:                                                       # 64 items x 32 lanes: more than one workgroup
        bad = build_gather_plan(out_row, tab_row, n_out, n_tab, w, chunk, group_slots=64).to(DEV)
        with pytest.raises(Exception):
            ops.gather_sum(bad, table.to(DEV))


def test_gather_sum_strided_views_and_determinism(ops):
    from tip_amd.plan import build_gather_plan
    g = torch.Generator().manual_seed(5)
    n, E, d = 64, 20000, 32
    out_row = torch.randint(0, n, (E,), generator=g)
    tab_row = torch.randint(0, n, (E,), generator=g)
    wide = torch.randn(n, 96, generator=g).to(DEV)
    plan = build_gather_plan(out_row, tab_row, n, n, None, 64).to(DEV)
    got = ops.gather_sum(plan, wide[:, 32:64])                        # column slice: ld = 96
    want = O.gather_sum(wide[:, 32:64].cpu().double(), tab_row, out_row, n)
    close(got, want)
    again = ops.gather_sum(plan, wide[:, 32:64])
    assert torch.equal(got, again)                                    # bitwise reproducible


# ------------------------------------------------------------------ gemm
@pytest.mark.parametrize('m,n,k', [(1, 1, 1), (33, 65, 17), (645, 32, 64), (130, 16, 300), (64, 64, 64), (7, 200, 5)])
def test_gemm_layouts(ops, m, n, k):
    g = torch.Generator().manual_seed(m + n + k)
    a = torch.randn(m, k, generator=g)
    b = torch.randn(k, n, generator=g)
    c0 = torch.randn(m, n, generator=g)
    want = a.double() @ b.double()
    ad, bd = a.to(DEV), b.to(DEV)
    close(ops.gemm(ad, bd), want, rtol=1e-5)
    close(ops.gemm(ad.t().contiguous().t(), bd), want, rtol=1e-5)                 # A column-major
    close(ops.gemm(ad, bd.t().contiguous().t()), want, rtol=1e-5)                 # B column-major
    close(ops.gemm(ad.t().contiguous().t(), bd.t().contiguous().t()), want, rtol=1e-5)
    close(ops.gemm(ad, bd, c_in=c0.to(DEV), relu=True, alpha=0.5), torch.relu(0.5 * want + c0.double()), rtol=1e-5)
    if k >= 16:
        close(ops.gemm(ad, bd, ksplit=3), want, rtol=1e-5)


def test_gemm_batched_and_reduce(ops):
    g = torch.Generator().manual_seed(1)
    a = torch.randn(5, 40, 24, generator=g)
    b = torch.randn(5, 24, 36, generator=g)
    x = torch.randn(40, 24, generator=g)
    ad, bd, xd = a.to(DEV), b.to(DEV), x.to(DEV)
    close(ops.gemm(ad, bd), torch.bmm(a.double(), b.double()), rtol=1e-5)
    close(ops.gemm(xd, bd), torch.einsum('mk,zkn->zmn', x.double(), b.double()), rtol=1e-5)      # shared A
    close(ops.gemm(ad, bd, reduce_batch=True), torch.einsum('zmk,zkn->mn', a.double(), b.double()), rtol=1e-5)
    bt = torch.randn(5, 36, 24, generator=g)                                                    # B given transposed
    close(ops.gemm(ad, bt.to(DEV).transpose(1, 2), reduce_batch=True),
          torch.einsum('zmk,znk->mn', a.double(), bt.double()), rtol=1e-5)
    # in-place accumulate (C aliases C_in) and column-slice output
    acc = torch.randn(40, 36, generator=g).to(DEV)
    want = acc.cpu().double() + torch.einsum('zmk,zkn->mn', a.double(), b.double())
    ops.gemm(ad, bd, out=acc, c_in=acc, reduce_batch=True)
    close(acc, want, rtol=1e-5)
    wide = torch.zeros(40, 100, device=DEV)
    ops.gemm(ad[0], bd[0], out=wide[:, 50:86])
    close(wide[:, 50:86], a[0].double() @ b[0].double(), rtol=1e-5)
    assert float(wide[:, :50].abs().max()) == 0 and float(wide[:, 86:].abs().max()) == 0


def test_gemm_is_exact_fp32_fma_chain(ops):
    """v_mfma_f32_32x32x2_f32 == k-ordered fmaf chain: integer-valued data must be exact."""
    g = torch.Generator().manual_seed(9)
    a = torch.randint(-8, 9, (70, 90), generator=g).float()
    b = torch.randint(-8, 9, (90, 45), generator=g).float()
    got = ops.gemm(a.to(DEV), b.to(DEV)).cpu()
    assert torch.equal(got, a @ b)
    # asymmetric B with A = I catches a transposed C/D map
    eye = torch.eye(45)
    assert torch.equal(ops.gemm(eye.to(DEV), b[:45].contiguous().to(DEV)).cpu(), b[:45])


@pytest.mark.parametrize('kind,m,n,k,ks', [
    ('thin_k', 1097, 4100, 32, None), ('thin_k', 300, 5000, 20, None), ('thin_k', 257, 4097, 1, None),
    ('thin_m', 32, 40000, 1097, 4), ('thin_m', 20, 33000, 300, 3), ('thin_m', 32, 33000, 257, None),
    ('kk', 1097, 32, 20640, 57), ('kk', 500, 17, 4100, 5), ('kk', 260, 32, 4096, None)])
def test_gemm_streaming_kernels(ops, kind, m, n, k, ks, monkeypatch):
    """The three wave-level streaming products (tipk_gemm.hip): Y = att.XB, dXB = att^T.dY, datt = dY.XB^T
    shapes incl. ragged tiles / k tails; the generic LDS-tiled kernel (option gemm_no_stream) is the cross-check."""
    import os
    g = torch.Generator().manual_seed(m + n + k)
    a = torch.randn(m, k, generator=g)
    b = torch.randn(k, n, generator=g)
    want = a.double() @ b.double()
    ad, bd = a.to(DEV), b.to(DEV)
    from tip_amd import _lib
    _lib.set_option('gemm_stream_kk', 1 if kind == 'kk' else 0)        # off by default (the tiled kernel is as fast)
    if kind == 'thin_m':
        ad = ad.t().contiguous().t()                                   # att^T: a view with a_sm = 1
    if kind == 'kk':
        bd = bd.t().contiguous().t()                                   # XB^T: k-contiguous B
    got = ops.gemm(ad, bd, ksplit=ks)
    close(got, want, rtol=2e-5, atol=1e-3)
    _lib.set_option('gemm_no_stream', 1)
    try:
        ref = ops.gemm(ad, bd, ksplit=ks)
    finally:
        _lib.set_option('gemm_no_stream', 0)
    close(ref, want, rtol=2e-5, atol=1e-3)
    if kind == 'thin_m':
        assert torch.equal(got, ref)                                   # same k order as the tiled kernel
    # epilogue (alpha, c_in, relu) and membership in a grouped launch
    if ks is None:
        c0 = torch.randn(m, n, generator=g).to(DEV)
        close(ops.gemm(ad, bd, c_in=c0, relu=True, alpha=0.5), torch.relu(0.5 * want + c0.cpu().double()), rtol=2e-5, atol=1e-3)
    x = torch.randn(70, 40, generator=g).to(DEV)
    y = torch.randn(40, 50, generator=g).to(DEV)
    outs = ops.gemm_group([ops.gemm_job(x, y), ops.gemm_job(ad, bd, ksplit=ks), ops.gemm_job(y.t(), x.t())])
    assert torch.equal(outs[0], ops.gemm(x, y)) and torch.equal(outs[2], ops.gemm(y.t(), x.t()))
    if kind == 'thin_m':
        assert torch.equal(outs[1], got)
    else:                                                              # grouped members use other bodies / k orders
        close(outs[1], want, rtol=2e-5, atol=1e-3)
    # exact on integer data (any k pairing is exact)
    ai = torch.randint(-4, 5, (m, k), generator=g).float()
    bi = torch.randint(-4, 5, (k, n), generator=g).float()
    aid, bid = ai.to(DEV), bi.to(DEV)
    if kind == 'thin_m':
        aid = aid.t().contiguous().t()
    if kind == 'kk':
        bid = bid.t().contiguous().t()
    assert torch.equal(ops.gemm(aid, bid, ksplit=ks).cpu(), ai @ bi)
    _lib.set_option('gemm_stream_kk', 0)


@pytest.mark.parametrize('r,nc,nb', [(1097, 20640, 32), (1097, 10320, 32), (70, 645 * 4, 7), (33, 513, 32), (5, 31, 1),
                                     (40, 70000, 32)])
def test_dy_products_fused(ops, r, nc, nb, monkeypatch):
    """tipk_rgcn_dy_products: d att = dY XB^T and d XB = att^T dY from one pass over dY, ragged row
    ranges / column chunks / base counts; reproducible; the two-GEMM path is the cross-check."""
    g = torch.Generator().manual_seed(r + nc + nb)
    gy = torch.randn(r, nc, generator=g)
    att = torch.randn(r, nb, generator=g)
    xb = torch.randn(nb, nc, generator=g)
    want_att = gy.double() @ xb.double().t()
    want_xb = att.double().t() @ gy.double()
    gyd, attd, xbd = gy.to(DEV), att.to(DEV), xb.to(DEV)
    g_att, g_xb = ops.dy_products(gyd, attd, xbd)
    tol = dict(rtol=2e-5, atol=2e-5 * float(want_att.abs().max()))
    close(g_att, want_att, **tol)
    close(g_xb, want_xb, rtol=2e-5, atol=2e-5 * float(want_xb.abs().max()))
    again = ops.dy_products(gyd, attd, xbd)
    assert torch.equal(again[0], g_att) and torch.equal(again[1], g_xb)
    monkeypatch.setenv('TIPK_NO_DY_FUSED', '1')
    ref_att, ref_xb = ops.dy_products(gyd, attd, xbd)
    close(ref_att, want_att, **tol)
    close(ref_xb, want_xb, rtol=2e-5, atol=2e-5 * float(want_xb.abs().max()))
    # exact on integers (every summation order is exact)
    gi = torch.randint(-3, 4, (r, nc), generator=g).float()
    ai = torch.randint(-3, 4, (r, nb), generator=g).float()
    xi = torch.randint(-3, 4, (nb, nc), generator=g).float()
    monkeypatch.delenv('TIPK_NO_DY_FUSED')
    e_att, e_xb = ops.dy_products(gi.to(DEV), ai.to(DEV), xi.to(DEV))
    assert torch.equal(e_att.cpu(), gi @ xi.t()) and torch.equal(e_xb.cpu(), ai.t() @ gi)


@pytest.mark.parametrize('R,N,d,nb', [(1097, 645, 32, 32), (1097, 645, 16, 32), (70, 100, 64, 7), (33, 129, 16, 32), (5, 31, 128, 1),
                                      (200, 77, 32, 20)])
def test_node_products_compact(ops, R, N, d, nb):
    """tipk_rgcn_node_products (tipk.h section 2d): the transposed gather writes dY compact and node-major, both
    products run on that form: == the dense products of the full dY (fp64), complete d XB (nodes without edges: zero
    blocks), reproducible, exact on integers; sources concentrated on a third of the nodes, relations without edges."""
    from tip_amd.plan import build_stream_plan
    g = torch.Generator().manual_seed(R + N + d)
    E = 30 * R
    rel = torch.randint(0, R, (E,), generator=g)
    rel[rel == R // 2] = 0                                                 # a relation without edges
    src = torch.randint(0, max(1, N // 3), (E,), generator=g)
    src[:E // 10] = torch.randint(0, N, (E // 10,), generator=g)
    dst = torch.randint(0, N, (E,), generator=g)
    split = ops.rel_stream_split(N, d)
    assert ops.node_products_slabs(N, d, R, nb) > 0
    sp = build_stream_plan(src, dst, rel, N, R, 16, d // split // 4, ops.rel_stream_piece(), compact=True).to(DEV)
    cr = sp.compact
    gp = torch.randn(N, d, generator=g)
    scale = torch.rand(N, generator=g) + 0.5
    att = torch.randn(R, nb, generator=g)
    xb = torch.randn(nb, N, d, generator=g)
    dy = O.gather_sum(gp.double() * scale.double().unsqueeze(1), dst, rel * N + src, R * N).view(R, N * d)
    want_att = dy @ xb.double().view(nb, N * d).t()
    want_xb = (att.double().t() @ dy).view(nb, N, d)
    dyc = ops.rel_stream_bwd(sp, gp.to(DEV), row_scale=scale.to(DEV))
    assert dyc.shape == (cr.n_rows + 1, d) and not bool(dyc[-1].any())
    rows = (cr.pos.long()[:, :R].t().reshape(-1))                          # (r, u) -> compact row | n_rows
    close(dyc[rows.to(DEV)].view(R, N * d), dy)
    job, g_xb = ops.node_products(dyc, cr, att.to(DEV), xb.to(DEV))
    ops.gemm_group([], [job])
    close(job.out, want_att, rtol=2e-5, atol=2e-5 * float(want_att.abs().max()))
    close(g_xb, want_xb, rtol=2e-5, atol=2e-5 * float(want_xb.abs().max()))
    job2, g_xb2 = ops.node_products(dyc, cr, att.to(DEV), xb.to(DEV))
    ops.gemm_group([], [job2])
    assert torch.equal(job2.out, job.out) and torch.equal(g_xb2, g_xb)
    # XB handed over a second time as [N, d, bases] (the d att product reads it coalesced): the same sums in the same order
    job3, g_xb3 = ops.node_products(dyc, cr, att.to(DEV), xb.to(DEV), xb.permute(1, 2, 0).contiguous().to(DEV))
    ops.gemm_group([], [job3])
    assert torch.equal(job3.out, job.out) and torch.equal(g_xb3, g_xb)
    # exact on integers
    gi = torch.randint(-3, 4, (N, d), generator=g).float()
    ai = torch.randint(-3, 4, (R, nb), generator=g).float()
    xi = torch.randint(-3, 4, (nb, N, d), generator=g).float()
    dyi = O.gather_sum(gi.double(), dst, rel * N + src, R * N).view(R, N * d)
    dyc = ops.rel_stream_bwd(sp, gi.to(DEV))
    job, g_xb = ops.node_products(dyc, cr, ai.to(DEV), xi.to(DEV))
    ops.gemm_group([], [job])
    assert torch.equal(job.out.cpu().double(), dyi @ xi.double().view(nb, N * d).t())
    assert torch.equal(g_xb.cpu().double(), (ai.double().t() @ dyi).view(nb, N, d))


def _random_dd_graph(N, R, per_rel, g, symmetric, hub=True):
    """directed edge list of R relations over N nodes: (src, dst, rel); symmetric -> every pair in both directions;
    a hub node linked to most others, nodes without any edge, a relation with one pair, duplicate-free inside a relation."""
    src, dst, rel = [], [], []
    for r in range(R):
        m = 1 if r == R - 1 else int(torch.randint(1, per_rel, (1,), generator=g))
        u = torch.randint(0, max(2, (2 * N) // 3), (m,), generator=g)
        v = torch.randint(0, max(2, (2 * N) // 3), (m,), generator=g)
        if hub and r % 3 == 0:
            u[: m // 2] = 1
        k = u != v
        u, v = u[k], v[k]
        if symmetric:
            key = torch.unique(torch.minimum(u, v) * N + torch.maximum(u, v))
            a, b = key // N, key % N
            u, v = torch.cat([a, b]), torch.cat([b, a])
        else:
            key = torch.unique(u * N + v)
            u, v = key // N, key % N
        src.append(u); dst.append(v); rel.append(torch.full((u.numel(),), r))
    return torch.cat(src), torch.cat(dst), torch.cat(rel)


@pytest.mark.parametrize('R,N,d,symmetric,per_rel', [(1097, 645, 32, True, 1200), (1097, 645, 16, True, 1200), (40, 100, 32, False, 300),
                                                     (33, 129, 16, False, 200), (7, 31, 32, True, 60), (200, 77, 16, True, 90)])
def test_pair_grads_backward(ops, R, N, d, symmetric, per_rel):
    """tipk_rgcn_pair_grads + tipk_stream_gather_parts (tipk.h section 2e): the backward pass of the pair form from the cells
    the forward pass left (half of them on a symmetric graph) == the dense definition in fp64 (d XB complete, d att over
    every directed edge), reproducible bit for bit, exact on integers; hub node, nodes without edges, one-pair relation."""
    from tip_amd.plan import build_pair_bwd_plan
    nb = 32
    g = torch.Generator().manual_seed(R + N + d)
    src, dst, rel = _random_dd_graph(N, R, per_rel, g, symmetric)
    scale = 1.0 / torch.bincount(dst, minlength=N).clamp(min=1).float()
    plan = build_pair_bwd_plan(src, dst, rel, N, R, scale, symmetric, 32, nb // 4, ops.rel_stream_piece()).to(DEV)
    n_pad = -(-N // 8) * 8

    def run(att, xb, gz):
        # the buffers as the forward pass leaves them: cells C[u, v, :] (u <= v only on a symmetric graph) + trailing zeros,
        # XB node-major with rows padded to 32 columns
        C = torch.zeros(N * N, nb, dtype=torch.float64)
        C.index_add_(0, src * N + dst, att.double()[rel])
        want_xb = torch.einsum('uvb,vc->buc', C.view(N, N, nb), gz.double() * scale.double().unsqueeze(1))
        want_att = torch.zeros(R, nb, dtype=torch.float64)
        want_att.index_add_(0, rel, torch.einsum('ebc,ec->eb', xb.double()[src], (gz.double() * scale.double().unsqueeze(1))[dst]))
        flat = torch.zeros(n_pad * N * nb + 64)
        cm = C.float().view(N, N, nb).clone()
        if symmetric:
            cm[~torch.triu(torch.ones(N, N, dtype=torch.bool))] = float('nan')     # never read: the mirrored half does not exist
        flat[:N * N * nb] = cm.view(-1)
        xb_pad = torch.zeros(n_pad, nb, 32)
        xb_pad[:N, :, :d] = xb
        flat, xb_pad = flat.to(DEV), xb_pad.to(DEV)
        cells = flat[:n_pad * N * nb].view(n_pad, N, nb)
        job, g_xb = ops.pair_backward(plan, cells, xb_pad[:, :, :d], gz.to(DEV))
        ops.gemm_group([], [job])
        job2, g_xb2 = ops.pair_backward(plan, cells, xb_pad[:, :, :d], gz.to(DEV))
        ops.gemm_group([], [job2])
        assert torch.equal(job2.out, job.out) and torch.equal(g_xb2, g_xb)
        return job.out, g_xb, want_att, want_xb

    att, xb, gz = torch.randn(R, nb, generator=g), torch.randn(N, nb, d, generator=g), torch.randn(N, d, generator=g)
    g_att, g_xb, want_att, want_xb = run(att, xb, gz)
    close(g_att, want_att, rtol=2e-5, atol=2e-5 * float(want_att.abs().max()))
    close(g_xb, want_xb, rtol=2e-5, atol=2e-5 * float(want_xb.abs().max()))
    # exact on integers (1 / deg = a power of two everywhere: every product and every summation order is exact)
    scale = torch.full((N,), 0.5)
    plan = build_pair_bwd_plan(src, dst, rel, N, R, scale, symmetric, 32, nb // 4, ops.rel_stream_piece()).to(DEV)
    ai = torch.randint(-3, 4, (R, nb), generator=g).float()
    xi = torch.randint(-3, 4, (N, nb, d), generator=g).float()
    gi = torch.randint(-3, 4, (N, d), generator=g).float()
    g_att, g_xb, want_att, want_xb = run(ai, xi, gi)
    assert torch.equal(g_att.cpu().double(), want_att) and torch.equal(g_xb.cpu().double(), want_xb)


@pytest.mark.parametrize('S,N,nb,d_out', [(81, 645, 32, 16), (5, 7, 32, 32), (40, 130, 8, 16), (1, 2, 3, 5)])
def test_sum_slabs_xb_layer_handover(ops, S, N, nb, d_out):
    """tipk_sum_slabs_xb (tipk.h section 2g): the slab sum that ends a layer + the next layer's XB / X root products in one
    launch == `sum_slabs` followed by the two products (fp64 references); the padding columns of the XB buffer stay
    untouched; odd row counts; bitwise repeat."""
    g = torch.Generator().manual_seed(S + N)
    slabs = torch.randn(S, N, 32, generator=g).to(DEV)
    scale = (torch.rand(N, generator=g) + 0.5).to(DEV)
    addend = torch.randn(N, 32, generator=g).to(DEV)
    basis = torch.randn(nb, 32, d_out, generator=g).to(DEV)
    root = torch.randn(32, d_out, generator=g).to(DEV)
    n_pad = -(-N // 8) * 8
    for relu in (True, False):
        xb_pad = torch.full((n_pad, nb, 32), 7.0, device=DEV)
        x = torch.empty(N, 32, device=DEV)
        xroot = ops.sum_slabs_xb(slabs, scale, addend, relu, x, basis, root, xb_pad)
        want_x = ops.sum_slabs(slabs, row_scale=scale, addend=addend, relu=relu)
        close(x, want_x.cpu(), rtol=1e-6, atol=1e-6)                       # (same lanes and order at 81 slabs; the compiler may fuse the epilogue differently)
        ref = (slabs.double().sum(0) * scale.double().unsqueeze(1) + addend.double()).cpu()
        close(x, torch.relu(ref) if relu else ref, rtol=2e-5, atol=2e-5)
        want_r = x.double().cpu() @ root.double().cpu()
        want_b = torch.einsum('ni,bio->nbo', x.double().cpu(), basis.double().cpu())
        close(xroot, want_r, rtol=2e-5, atol=2e-5 * float(want_r.abs().max()))
        close(xb_pad[:N, :, :d_out], want_b, rtol=2e-5, atol=2e-5 * float(want_b.abs().max()))
        assert bool((xb_pad[:N, :, d_out:] == 7.0).all()) and bool((xb_pad[N:] == 7.0).all())
        xb2 = torch.full((n_pad, nb, 32), 7.0, device=DEV)
        x2 = torch.empty(N, 32, device=DEV)
        assert torch.equal(ops.sum_slabs_xb(slabs, scale, addend, relu, x2, basis, root, xb2), xroot) and torch.equal(xb2, xb_pad)


@pytest.mark.parametrize('R,N,d_in,nb,E', [(200, 1000, 128, 32, 300000), (37, 50, 40, 7, 3000), (5, 31, 128, 32, 40), (2000, 300, 64, 32, 90000)])
def test_dest_products_forward(ops, R, N, d_in, nb, E):
    """tipk_rgcn_dest_products (tipk.h section 2f): T[b, v, :] = sum over the edges into v of att[r_e, b] x[src_e, :] == the
    definition in fp64; together with the batch-reduced product over the bases == sum_r A_r X W_r (the aggregate the Y route
    gathers); nodes without incoming edges, degrees that are no multiple of 32, bitwise repeat, exact on integers."""
    from tip_amd.plan import build_dest_plan
    g = torch.Generator().manual_seed(R + N + d_in)
    rel = torch.sort(torch.randint(0, R, (E,), generator=g)).values
    src = torch.randint(0, N, (E,), generator=g)
    dst = torch.randint(0, max(1, (3 * N) // 4), (E,), generator=g)          # a quarter of the nodes receive nothing
    dst[: E // 5] = 1                                                      # a hub
    bits = ops.dest_products_bits(N, R, nb, d_in)
    assert bits > 0
    dp = build_dest_plan(src, dst, rel, N, R, bits).to(DEV)

    def run(x, att):
        t = ops.dest_products(dp, x.to(DEV), att.to(DEV))
        assert torch.equal(t, ops.dest_products(dp, x.to(DEV), att.to(DEV)))
        want = torch.zeros(N, nb, d_in, dtype=torch.float64)
        want.index_add_(0, dst, att.double()[rel].unsqueeze(2) * x.double()[src].unsqueeze(1))
        return t, want.permute(1, 0, 2)
    x, att = torch.randn(N, d_in, generator=g), torch.randn(R, nb, generator=g)
    t, want = run(x, att)
    close(t, want, rtol=2e-5, atol=2e-5 * float(want.abs().max()))
    basis = torch.randn(nb, d_in, 16, generator=g)
    agg = ops.gemm(t, basis.to(DEV), reduce_batch=True)
    w_r = torch.einsum('rb,bio->rio', att.double(), basis.double())
    ref = torch.zeros(N, 16, dtype=torch.float64)
    ref.index_add_(0, dst, torch.einsum('ei,eio->eo', x.double()[src], w_r[rel]))
    close(agg, ref, rtol=2e-5, atol=2e-5 * float(ref.abs().max()))
    xi, ai = torch.randint(-3, 4, (N, d_in), generator=g).float(), torch.randint(-3, 4, (R, nb), generator=g).float()
    t, want = run(xi, ai)
    assert torch.equal(t.cpu().double(), want)


@pytest.mark.parametrize('R,N,ch,nb,E', [(70, 50, 32, 32, 4000), (33, 37, 64, 8, 900), (129, 21, 128, 32, 6000), (5, 16, 32, 3, 40)])
def test_row_products_both_passes(ops, R, N, ch, nb, E):
    """tipk_rgcn_row_products (tipk.h section 2h): the (relation, node) row sums assembled in LDS and multiplied there ==
    the definition in fp64 -- T[b, v, :] = sum_r att[r, b] S[(r, v), :] (forward: with the batch-reduced product over the
    bases == sum_r A_r X W_r) and, with XB, d att[r, b] = sum_v <S[(r, v)], XB[b, v]> (the transposed pass's two products of
    dY).  Nodes without edges, a hub, relations without edges, R no multiple of 32, node counts no multiple of 16; bitwise
    repeat; exact on integers; the plan interpreted in torch gives the same."""
    from tip_amd.plan import build_row_stream_plan, execute_row_stream_reference
    assert ops.row_products_supported(N, R, nb, ch)
    g = torch.Generator().manual_seed(R + N + ch)
    rel = torch.randint(0, R, (E,), generator=g)
    rel[rel == R // 2] = 0                                                 # a relation without edges
    key = torch.randint(0, max(1, (3 * N) // 4), (E,), generator=g)          # a quarter of the nodes own no row
    key[: E // 5] = 1                                                      # a hub
    other = torch.randint(0, N, (E,), generator=g)
    rp = build_row_stream_plan(key, other, rel, N, R).to(DEV)

    def want(table, att, xb):
        s = torch.zeros(R * N, ch, dtype=torch.float64)
        s.index_add_(0, rel * N + key, table.double()[other])
        s = s.view(R, N, ch)
        return torch.einsum('rb,rvc->bvc', att.double(), s), torch.einsum('rvc,bvc->rb', s, xb.double())

    def run(table, att, xb):
        t1 = ops.row_products(rp, table.to(DEV), att.to(DEV))
        job, t2 = ops.row_products(rp, table.to(DEV), att.to(DEV), xb.to(DEV).view(nb, N * ch))
        ops.gemm_group([], [job])
        job_b, t3 = ops.row_products(rp, table.to(DEV), att.to(DEV), xb.to(DEV).view(nb, N * ch))
        ops.gemm_group([], [job_b])
        assert torch.equal(t1, t2) and torch.equal(t2, t3) and torch.equal(job.out, job_b.out)
        return t1, job.out
    table, att, xb = torch.randn(N, ch, generator=g), torch.randn(R, nb, generator=g), torch.randn(nb, N, ch, generator=g)
    t, datt = run(table, att, xb)
    wt, wa = want(table, att, xb)
    close(t, wt, rtol=2e-5, atol=2e-5 * float(wt.abs().max()))
    close(datt, wa, rtol=2e-5, atol=2e-5 * float(wa.abs().max()))
    rt, ra = execute_row_stream_reference(rp, table, att, xb)
    close(rt, wt, rtol=1e-12, atol=1e-12)
    close(ra, wa, rtol=1e-12, atol=1e-10)
    basis = torch.randn(nb, ch, 16, generator=g)
    agg = ops.gemm(t, basis.to(DEV), reduce_batch=True)
    w_r = torch.einsum('rb,bio->rio', att.double(), basis.double())
    ref = torch.zeros(N, 16, dtype=torch.float64)
    ref.index_add_(0, key, torch.einsum('ei,eio->eo', table.double()[other], w_r[rel]))
    close(agg, ref, rtol=2e-5, atol=2e-5 * float(ref.abs().max()))
    ti, ai = torch.randint(-3, 4, (N, ch), generator=g).float(), torch.randint(-3, 4, (R, nb), generator=g).float()
    xi = torch.randint(-2, 3, (nb, N, ch), generator=g).float()
    t, datt = run(ti, ai, xi)
    wt, wa = want(ti, ai, xi)
    assert torch.equal(t.cpu().double(), wt) and torch.equal(datt.cpu().double(), wa)


@pytest.mark.parametrize('R,N,ch,nb,E', [(70, 50, 64, 32, 4000), (33, 37, 128, 8, 900), (129, 21, 192, 32, 6000), (5, 16, 64, 3, 40)])
def test_row_products_s_both_passes(ops, R, N, ch, nb, E):
    """tipk_rgcn_row_products_s (wave-uniform entries: scalar entry words, buffer loads with a scalar offset, LDS stores
    through M0) == the definition in fp64 and == `tipk_rgcn_row_products` on the same graph up to summation order; nodes
    without edges, a hub, a relation without edges, R no multiple of 32, node counts no multiple of 8; bitwise repeat; exact
    on integers."""
    from tip_amd.plan import build_row_stream_plan_s, build_row_stream_plan
    assert ops.row_products_s_supported(N, R, nb, ch)
    g = torch.Generator().manual_seed(R + N + ch)
    rel = torch.randint(0, R, (E,), generator=g)
    rel[rel == R // 2] = 0
    key = torch.randint(0, max(1, (3 * N) // 4), (E,), generator=g)
    key[: E // 5] = 1
    other = torch.randint(0, N, (E,), generator=g)
    rp = build_row_stream_plan_s(key, other, rel, N, R).to(DEV)
    rv = build_row_stream_plan(key, other, rel, N, R).to(DEV)

    def want(table, att, xb):
        s = torch.zeros(R * N, ch, dtype=torch.float64)
        s.index_add_(0, rel * N + key, table.double()[other])
        s = s.view(R, N, ch)
        return torch.einsum('rb,rvc->bvc', att.double(), s), torch.einsum('rvc,bvc->rb', s, xb.double())

    def run(plan, table, att, xb):
        t1 = ops.row_products(plan, table.to(DEV), att.to(DEV))
        job, t2 = ops.row_products(plan, table.to(DEV), att.to(DEV), xb.to(DEV).view(nb, N * ch))
        ops.gemm_group([], [job])
        job_b, t3 = ops.row_products(plan, table.to(DEV), att.to(DEV), xb.to(DEV).view(nb, N * ch))
        ops.gemm_group([], [job_b])
        assert torch.equal(t1, t2) and torch.equal(t2, t3) and torch.equal(job.out, job_b.out)
        return t1, job.out
    table, att, xb = torch.randn(N, ch, generator=g), torch.randn(R, nb, generator=g), torch.randn(nb, N, ch, generator=g)
    t, datt = run(rp, table, att, xb)
    wt, wa = want(table, att, xb)
    close(t, wt, rtol=2e-5, atol=2e-5 * float(wt.abs().max()))
    close(datt, wa, rtol=2e-5, atol=2e-5 * float(wa.abs().max()))
    tv, dv = run(rv, table, att, xb)
    close(t, tv.cpu().double(), rtol=2e-5, atol=2e-5 * float(wt.abs().max()))
    close(datt, dv.cpu().double(), rtol=2e-5, atol=2e-5 * float(wa.abs().max()))
    ti, ai = torch.randint(-3, 4, (N, ch), generator=g).float(), torch.randint(-3, 4, (R, nb), generator=g).float()
    xi = torch.randint(-2, 3, (nb, N, ch), generator=g).float()
    t, datt = run(rp, ti, ai, xi)
    wt, wa = want(ti, ai, xi)
    assert torch.equal(t.cpu().double(), wt) and torch.equal(datt.cpu().double(), wa)
    # a table whose rows are longer than its channels (a view): the scalar offsets follow the stride
    wide = torch.randn(N, ch + 64, generator=g)
    t = ops.row_products(rp, wide.to(DEV)[:, :ch], att.to(DEV))
    wt, _ = want(wide[:, :ch], att, xb)
    close(t, wt, rtol=2e-5, atol=2e-5 * float(wt.abs().max()))


@pytest.mark.parametrize('R,N,d', [(70, 645, 32), (33, 100, 16), (1097, 37, 8)])
def test_unwritten_rows_masked_end_to_end(ops, R, N, d):
    """Rows (relation, node) without edges: the wave-stream gather with write_zeros=False leaves them untouched
    and `dy_products(row_used=...)` clears whatever the buffer holds there (NaN bit patterns included): same
    d att / d XB, bit for bit, as the zero-filled run."""
    from tip_amd.plan import build_stream_plan
    g = torch.Generator().manual_seed(R + N)
    E = 40 * R
    rel = torch.randint(0, R, (E,), generator=g)
    src = torch.randint(0, max(1, N // 3), (E,), generator=g)              # two thirds of the nodes never a source
    dst = torch.randint(0, N, (E,), generator=g)
    split = ops.rel_stream_split(N, d)
    sp = build_stream_plan(src, dst, rel, N, R, 16, d // split // 4, ops.rel_stream_piece()).to(DEV)
    cnt = torch.bincount(rel * N + src, minlength=R * N).view(R, N)
    bits = sp.row_used.cpu().long() & 0xffffffff
    for r in (0, R // 2, R - 1):
        assert torch.equal(((bits[r >> 5] >> (r & 31)) & 1).bool(), cnt[r] > 0)
    gp = torch.randn(N, d, generator=g).to(DEV)
    full = ops.rel_stream_bwd(sp, gp)
    # poison the allocation the next call will get, then gather without the zero rows
    poison = torch.full((R * N, d), float('nan'), device=DEV)
    ptr = poison.data_ptr()
    del poison
    part = ops.rel_stream_bwd(sp, gp, write_zeros=False)
    if part.data_ptr() == ptr:                                             # the caching allocator handed the block back
        assert bool(torch.isnan(part).any()) and bool(torch.isnan(part.view(R, N, d)[cnt.to(DEV) == 0]).all())
    assert torch.equal(part.view(R, N, d)[cnt.to(DEV) > 0], full.view(R, N, d)[cnt.to(DEV) > 0])
    nb = 32
    att = torch.randn(R, nb, generator=g).to(DEV)
    xb = torch.randn(nb, N * d, generator=g).to(DEV)
    assert ops.dy_products_fused(R, N * d, nb)
    want = ops.dy_products(full.view(R, N * d), att, xb)
    got = ops.dy_products(part.view(R, N * d), att, xb, sp.row_used, N)
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    assert not bool(torch.isnan(got[0]).any()) and not bool(torch.isnan(got[1]).any())


def test_gemm_output_beyond_4GB(ops):
    """Y of the synthetic config is 10 GB: rows past the 4 GB mark must land where they belong
    (32-bit offsets are tile-local only)."""
    g = torch.Generator().manual_seed(3)
    m, k, n = 2100, 32, 600000                                         # C = 5.04 GB
    a = torch.randn(m, k, generator=g).to(DEV)
    b = torch.randn(k, n, generator=g).to(DEV)
    c = ops.gemm(a, b)
    for r in (0, 1, 1000, 1789, 1790, 2099):                           # 1790 * 600000 * 4 B > 2^32
        want = (a[r].double() @ b.double()).cpu()
        close(c[r], want, rtol=2e-5, atol=1e-4)
    del c
    torch.cuda.empty_cache()


def test_gemm_group_bit_identical_to_single_launches(ops):
    """tipk_gemm_f32_group / tipk_sum_slabs_group: the R-GCN backward's mix of shapes (batched, split-K,
    batch-reduced on top of another member's output, transposed views) in one launch == one by one."""
    g = torch.Generator().manual_seed(77)
    n, d_in, d_out, nb = 645, 64, 32, 8
    x = torch.randn(n, d_in, generator=g).to(DEV)
    gr = torch.randn(n, d_out, generator=g).to(DEV)
    g_xb = torch.randn(nb, n, d_out, generator=g).to(DEV)
    basis = torch.randn(nb, d_in, d_out, generator=g).to(DEV)
    root = torch.randn(d_in, d_out, generator=g).to(DEV)

    def build(gx_buf):
        j_basis = ops.gemm_job(x.t(), g_xb)
        j_root = ops.gemm_job(x.t(), gr)
        j_xr = ops.gemm_job(gr, root.t(), out=gx_buf, ksplit=1)      # its output is accumulated onto by j_xq
        j_xq = ops.gemm_job(g_xb, basis.transpose(1, 2), out=gx_buf, c_in=gx_buf, reduce_batch=True)
        return [j_basis, j_root, j_xr, j_xq]

    jobs = build(torch.empty(n, d_in, device=DEV))
    assert jobs[0].slabs is not None and jobs[1].slabs is not None and jobs[3].slabs is not None
    got = [o.clone() for o in ops.gemm_group(jobs)]
    want_basis = ops.gemm(x.t(), g_xb)
    want_root = ops.gemm(x.t(), gr)
    want_x = ops.gemm(gr, root.t())
    want_x = ops.gemm(g_xb, basis.transpose(1, 2), out=want_x, c_in=want_x, reduce_batch=True)
    assert torch.equal(got[0], want_basis) and torch.equal(got[1], want_root)
    assert torch.equal(got[2], want_x) and jobs[3].out.data_ptr() == jobs[2].out.data_ptr()
    close(got[0], torch.einsum('nk,bno->bko', x.double().cpu(), g_xb.double().cpu()), rtol=1e-5, atol=1e-4)
    close(got[2], gr.double().cpu() @ root.double().cpu().t()
          + torch.einsum('bno,bko->nk', g_xb.double().cpu(), basis.double().cpu()), rtol=1e-5, atol=1e-4)
    # more members than TIPK_GROUP_MAX are chunked; empty group is a no-op; extra slab jobs ride along
    many = [ops.gemm_job(x.t(), gr) for _ in range(8)]
    part = torch.randn(5, 40, 16, generator=g).to(DEV)
    sj = ops.slab_job(part, alpha=0.5, relu=True)
    outs = ops.gemm_group(many, [sj])
    assert all(torch.equal(o, want_root) for o in outs)
    assert torch.equal(sj.out, ops.sum_slabs(part, alpha=0.5, relu=True))
    assert ops.gemm_group([]) == []


@pytest.mark.parametrize('n,d_in,d_out,nb', [(645, 64, 32, 32), (645, 32, 16, 32), (37, 10, 7, 3), (1000, 48, 40, 5)])  # (<= 64 K tiles)
def test_wg_gemm_group_vs_fp64_exact_on_integers_and_repeatable(ops, n, d_in, d_out, nb):
    """tipk_gemm_wg_group: the four products that end an R-GCN layer's backward pass (batched X^T dXB, X^T g, the
    batch-reduced dXB basis^T with g root^T added as a second product and the ReLU gate of the layer's input) and a slab
    sum in ONE launch, reductions split over the waves of one workgroup per output tile: vs fp64, exact on small
    integers (any order of the sum), bit-identical when repeated, ragged tiles and transposed views."""
    g = torch.Generator().manual_seed(5 + n)
    for integers in (False, True):
        def rnd(*shape):
            if integers:
                return torch.randint(-3, 4, shape, generator=g).float().to(DEV)
            return torch.randn(*shape, generator=g).to(DEV)
        x, gr, g_xb = rnd(n, d_in), rnd(n, d_out), rnd(nb, n, d_out)
        basis, root = rnd(nb, d_in, d_out), rnd(d_in, d_out)
        part = rnd(6, 50, 8)
        outs = []
        for rep in range(2):
            jb = ops.wg_gemm_job(x.t(), g_xb)
            jr = ops.wg_gemm_job(x.t(), gr)
            jx = ops.wg_gemm_job(g_xb, basis.transpose(1, 2), reduce_batch=True, a2=gr, b2=root.t(), gate=x, alpha=0.5)
            assert jb is not None and jr is not None and jx is not None
            sj = ops.slab_job(part, alpha=2.0)
            ops.wg_gemm_group([jb, jr, jx], [sj])
            outs.append([jb.out.clone(), jr.out.clone(), jx.out.clone(), sj.out.clone()])
        for a, b in zip(*outs):
            assert torch.equal(a, b)
        xd, gd, xbd, bd, rd = (t.double().cpu() for t in (x, gr, g_xb, basis, root))
        want = [torch.einsum('nk,bno->bko', xd, xbd), xd.t() @ gd,
                torch.where(xd > 0, 0.5 * (torch.einsum('bno,bko->nk', xbd, bd) + gd @ rd.t()), torch.zeros(())),
                2.0 * part.double().cpu().sum(0)]
        for got, w in zip(outs[0], want):
            if integers:
                assert torch.equal(got.double().cpu(), w)
            else:
                close(got, w, rtol=1e-5, atol=2e-4)
    assert torch.equal(outs[0][3], ops.sum_slabs(part, alpha=2.0))
    # a plain product with output strides (a column block of a wider buffer) and no second term; a long reduction is refused
    wide = torch.zeros(n, d_in + 8, device=DEV)
    j = ops.wg_gemm_job(gr, root.t(), out=wide[:, 8:])
    ops.wg_gemm_group([j])
    assert torch.equal(wide[:, :8], torch.zeros(n, 8, device=DEV))
    want_w = gr.double().cpu() @ root.double().cpu().t()
    assert torch.equal(wide[:, 8:].double().cpu(), want_w) if integers else True
    assert ops.wg_gemm_job(torch.zeros(4, 5000, device=DEV), torch.zeros(5000, 4, device=DEV)) is None


@pytest.mark.parametrize('n,d_in,d_out,n_rows', [(19081, 32, 16, 3640), (500, 16, 16, 77), (300, 64, 16, 300), (200, 32, 8, 50)])
def test_gcn_conv_aggregate_first_on_kept_rows(ops, n, d_in, d_out, n_rows):
    """`gather_sum_lin` / `ops.gcn_conv_agg_first`: a GCN layer for a subset of its rows, aggregation first and the dense
    map on the kept rows in the same launch -- output and all three gradients equal the transform-first layer restricted
    to the rows (fp64 reference through the dense normalised adjacency), bitwise repeatable."""
    from tip_amd.layers import gcn_norm_graph
    from tip_amd.plan import group_slots_for
    g = torch.Generator().manual_seed(n + d_in)
    e = 12 * n
    ei = torch.stack([torch.randint(0, n, (e,), generator=g), torch.randint(0, n, (e,), generator=g)])
    ei[1, : e // 20] = 3                                                   # a hub row: split into pieces
    rows = torch.sort(torch.randperm(n, generator=g)[:n_rows]).values
    if 3 not in rows.tolist():
        rows = torch.sort(torch.cat([rows[:-1], torch.tensor([3])])).values
    assert ops.gather_sum_lin_supported(d_in, d_out, group_slots_for(d_in))
    graph = gcn_norm_graph(ei.to(DEV), n, d=d_in, rows=rows.to(DEV))
    x = torch.randn(n, d_in, generator=g).to(DEV).requires_grad_()
    wt = torch.randn(d_in, d_out, generator=g).to(DEV)                     # the layers' storage: [in, out] behind a [out, in] view
    w = wt.t().requires_grad_()
    b = torch.randn(d_out, generator=g).to(DEV).requires_grad_()
    up = torch.randn(n_rows, d_out, generator=g).to(DEV)
    outs = []
    for rep in range(2):
        for t in (x, w, b):
            t.grad = None
        out = ops.gcn_conv_agg_first(x, w, b, graph, relu=True)
        out.backward(up)
        outs.append([out.detach().clone(), x.grad.clone(), w.grad.clone(), b.grad.clone()])
    for a, c in zip(*outs):
        assert torch.equal(a, c)
    # fp64 reference: dense D^-1/2 (A + I) D^-1/2 restricted to the kept rows
    row, col = ei[0], ei[1]
    keep = row != col
    row, col = torch.cat([row[keep], torch.arange(n)]), torch.cat([col[keep], torch.arange(n)])
    deg = torch.bincount(col, minlength=n).double()
    dis = deg.pow(-0.5)
    xd = x.detach().double().cpu().requires_grad_()
    wd = w.detach().double().cpu().requires_grad_()
    bd = b.detach().double().cpu().requires_grad_()
    xl = xd @ wd.t()
    msg = xl[row] * (dis[row] * dis[col]).unsqueeze(1)
    ref = torch.relu(torch.zeros(n, d_out, dtype=torch.float64).index_add_(0, col, msg)[rows] + bd)
    ref.backward(up.double().cpu())
    scale = float(ref.abs().max())
    close(outs[0][0], ref.detach(), rtol=1e-5, atol=1e-5 * scale)
    close(outs[0][1], xd.grad, rtol=1e-5, atol=1e-5 * float(xd.grad.abs().max()))
    close(outs[0][2], wd.grad, rtol=1e-5, atol=1e-5 * float(wd.grad.abs().max()))
    close(outs[0][3], bd.grad, rtol=1e-5, atol=1e-5 * float(bd.grad.abs().max()))
    assert outs[0][2].stride() == w.stride()                               # d W in the parameter's own layout
    # ReLU hand-over: x as the ReLU output of a producer -- dx comes back masked with (x > 0), the column sums of the masked dx
    # as partial rows in the link (what `_GCNConv(relu='gated_downstream')` turns into its bias gradient)
    if ops.gather_sum_epilogue_supported(graph.bwd, d_in):
        link = ops.GateLink()
        x2 = torch.relu(x.detach()).requires_grad_()
        out2 = ops.gcn_conv_agg_first(x2, w, b, graph, True, True, link)
        out2.backward(up)
        x3 = torch.relu(x.detach()).requires_grad_()
        ops.gcn_conv_agg_first(x3, w, b, graph, True).backward(up)
        masked = torch.where(x3.detach() > 0, x3.grad, torch.zeros_like(x3.grad))
        assert torch.equal(x2.grad, masked)
        close(link.parts.double().cpu().sum((0, 1)), masked.double().cpu().sum(0), rtol=1e-5, atol=1e-5 * float(masked.abs().sum(0).max()))


def test_gather_sum_with_slab_sums_riding_in_the_launch(ops):
    """tipk_gather_sum_riders: ordered slab sums as further workgroups of a grouped gather whose workgroups have 1024
    threads -- the gather's result and every sum equal the separate launches bit for bit; a plan with narrower workgroups
    (16-float rows) takes the riders as a launch of their own."""
    from tip_amd.plan import build_gather_plan, group_slots_for
    g = torch.Generator().manual_seed(21)
    n_out, n_tab = 3000, 700
    e = 40000
    dst, src = torch.randint(0, n_out, (e,), generator=g), torch.randint(0, n_tab, (e,), generator=g)
    dst[:5000] = 7                                                         # a hub row
    w = torch.rand(e, generator=g)
    parts = [torch.randn(256, 1, 32, generator=g).to(DEV), torch.randn(11, 32, 16, generator=g).to(DEV), torch.randn(57, 1, 16, generator=g).to(DEV)]
    for d in (32, 16):
        plan = build_gather_plan(dst, src, n_out, n_tab, w, 64, 'test', group_slots_for(d)).to(DEV)
        table = torch.randn(n_tab, d, generator=g).to(DEV)
        want = ops.gather_sum(plan, table)
        jobs = [ops.slab_job(p_, alpha=0.5) for p_ in parts]
        got = ops.gather_sum(plan, table, riders=jobs)
        assert torch.equal(got, want)
        for j, p_ in zip(jobs, parts):
            assert torch.equal(j.out, ops.sum_slabs(p_, alpha=0.5))
        assert bool(ops.lib().tipk_gather_sum_riders_supported(d, plan.group_slots)) == (d == 32)


def test_gather_sum_gate_and_column_sums_in_the_epilogue(ops):
    """tipk_gather_sum_riders with gate / colsum: out = gate > 0 ? sum : 0 and the per-workgroup column sums of those rows --
    == the plain gather masked afterwards (bit for bit) and its column sums (fp64), riders alongside, repeatable."""
    from tip_amd.plan import build_gather_plan, group_slots_for
    g = torch.Generator().manual_seed(31)
    n_out, n_tab, e, d = 19081, 3640, 300000, 32
    dst, src = torch.randint(0, n_out, (e,), generator=g), torch.randint(0, n_tab, (e,), generator=g)
    dst[:9000] = 11                                                        # a hub row
    w = torch.rand(e, generator=g)
    plan = build_gather_plan(dst, src, n_out, n_tab, w, 64, 'test', group_slots_for(d)).to(DEV)
    assert ops.gather_sum_epilogue_supported(plan, d)
    table = torch.randn(n_tab, d, generator=g).to(DEV)
    gate = torch.randn(n_out, d, generator=g).to(DEV)
    part = torch.randn(40, 1, 16, generator=g).to(DEV)
    plain = ops.gather_sum(plan, table)
    want = torch.where(gate > 0, plain, torch.zeros_like(plain))
    res = []
    for rep in range(2):
        job = ops.slab_job(part)
        out, parts = ops.gather_sum(plan, table, riders=[job], gate=gate, colsum=True)
        assert torch.equal(out, want) and torch.equal(job.out, ops.sum_slabs(part))
        assert parts.shape == (-(-plan.items.shape[0] // plan.group_slots), 1, d)
        close(parts.double().cpu().sum((0, 1)), want.double().cpu().sum(0), rtol=1e-5, atol=1e-4 * float(want.abs().sum(0).max()))
        res.append(parts.clone())
    assert torch.equal(res[0], res[1])
    only_gate = ops.gather_sum(plan, table, gate=gate)
    assert torch.equal(only_gate, want)


def test_gemm_reduce_batch_in_groups(ops):
    """sum_z a[z] @ b[z] with the terms summed in groups of `kgroup` (one slab per group): the pair-form D-D product."""
    g = torch.Generator().manual_seed(12)
    for z, m, k, n, kg in ((648, 645, 32, 32, 8), (24, 100, 16, 16, 4), (16, 33, 32, 8, 16)):
        a = torch.randn(z, m, k, generator=g).to(DEV)
        b = torch.randn(z, k, n, generator=g).to(DEV)
        job = ops.gemm_job(a, b, reduce_batch=True, kgroup=kg)
        assert job.slabs.shape == (z // kg, m, n)
        out = ops.gemm_group([job])[0]
        want = torch.einsum('zmk,zkn->mn', a.double().cpu(), b.double().cpu())
        close(out, want, rtol=2e-5, atol=2e-5 * float(want.abs().max()))
        per_group = torch.einsum('gzmk,gzkn->gmn', a.double().cpu().view(z // kg, kg, m, k), b.double().cpu().view(z // kg, kg, k, n))
        close(job.slabs, per_group, rtol=2e-5, atol=2e-5 * float(per_group.abs().max()))


@pytest.mark.parametrize('n,nb,d', [(645, 32, 32), (645, 32, 16), (100, 16, 8), (37, 8, 32), (70, 32, 5)])
def test_pair_product(ops, n, nb, d, monkeypatch):
    """tipk_pair_product (section 2c): slabs of sum_u cells[u] @ xb[u] per group of source nodes; the tiled GEMM
    with kbatch = group is the cross-check."""
    g = torch.Generator().manual_seed(n + nb + d)
    n_pad = -(-n // ops.PAIR_KGROUP) * ops.PAIR_KGROUP
    cells = torch.zeros(n_pad, n, nb)
    cells[:n] = torch.randn(n, n, nb, generator=g) * (torch.rand(n, n, 1, generator=g) < 0.3)
    xb = torch.zeros(n_pad, nb, d)
    xb[:n] = torch.randn(n, nb, d, generator=g)
    xb_host = xb
    xb_pad = torch.zeros(n_pad, nb, 32, device=DEV)                        # the kernel's layout: rows padded to 32 columns
    xb_pad[:, :, :d] = xb.to(DEV)

    class _Padded(object):                                                 # `xb.to(DEV)` below -> the padded device view
        def to(self, _):
            return xb_pad[:, :, :d]

        def double(self):
            return xb_host.double()
    xb = _Padded()
    slabs = ops.pair_product(cells.to(DEV), xb.to(DEV))
    assert slabs.shape == (n_pad // ops.PAIR_KGROUP, n, d)
    want = torch.einsum('guvb,gubc->gvc', cells.double().view(-1, ops.PAIR_KGROUP, n, nb), xb.double().view(-1, ops.PAIR_KGROUP, nb, d))
    close(slabs, want, rtol=2e-5, atol=2e-5 * float(want.abs().max()))
    assert torch.equal(slabs, ops.pair_product(cells.to(DEV), xb.to(DEV)))
    # XB written back with the bases innermost (the backward pass's operand), by the kernel and by the fallback
    want_t = xb.to(DEV)[:n].permute(0, 2, 1).contiguous()
    xbt = torch.full((n, d, nb), float('nan'), device=DEV)
    assert torch.equal(slabs, ops.pair_product(cells.to(DEV), xb.to(DEV), xbt=xbt)) and torch.equal(xbt, want_t)
    monkeypatch.setenv('TIPK_NO_PAIR_PRODUCT', '1')
    close(ops.pair_product(cells.to(DEV), xb.to(DEV)), want, rtol=2e-5, atol=2e-5 * float(want.abs().max()))
    xbt = torch.full((n, d, nb), float('nan'), device=DEV)
    ops.pair_product(cells.to(DEV), xb.to(DEV), xbt=xbt)
    assert torch.equal(xbt, want_t)
    monkeypatch.delenv('TIPK_NO_PAIR_PRODUCT')
    # symmetric cells: only the half with source <= destination is stored, the other half may hold anything
    sym = cells.clone()
    sym[:n] = torch.triu(cells[:n].permute(2, 0, 1)).permute(1, 2, 0)                 # keep u <= v
    full = sym.clone()
    full[:n] = sym[:n] + torch.triu(sym[:n].permute(2, 0, 1), 1).permute(2, 1, 0)      # mirror: C[v][u] = C[u][v]
    want_s = torch.einsum('guvb,gubc->gvc', full.double().view(-1, ops.PAIR_KGROUP, n, nb), xb.double().view(-1, ops.PAIR_KGROUP, nb, d))
    junk = sym.clone()
    lower = torch.tril(torch.ones(n, n), -1).bool()
    junk[:n][lower] = float('nan')
    got_s = ops.pair_product(junk.to(DEV), xb.to(DEV), symmetric=True)
    close(got_s, want_s, rtol=2e-5, atol=2e-5 * float(want_s.abs().max()))
    # cells of unlinked pairs are not fetched (`links`: bit r of word (u, t) = pair (u, 32 t + r) is linked): such cells may
    # hold anything -- the product reads zeros in their place -- and the result equals the product of the masked cells
    gl = torch.Generator().manual_seed(n + 1)
    n8, n32 = cells.shape[0], -(-n // 32) * 32
    link = torch.zeros(n8, n32, dtype=torch.bool)
    link[:n, :n] = torch.rand(n, n, generator=gl) < 0.08
    link[:n, :n] |= link[:n, :n].t().clone()                                         # symmetric links
    w = (link.view(n8, n32 // 32, 32).to(torch.int64) << torch.arange(32).view(1, 1, 32)).sum(-1)
    links = torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32).contiguous()
    keep = link[:, :n]                                                               # cell (u, v) is read iff the pair is linked
    masked = full * keep.unsqueeze(-1)
    want_l = torch.einsum('guvb,gubc->gvc', masked.double().view(-1, ops.PAIR_KGROUP, n, nb), xb.double().view(-1, ops.PAIR_KGROUP, nb, d))
    poison = sym.clone()
    never = ~(keep[:n] | keep[:n].t())                                               # (a stored cell serves (u, v) and (v, u))
    poison[:n][never] = float('nan')
    poison[:n][lower] = float('nan')
    got_l = ops.pair_product(poison.to(DEV), xb.to(DEV), symmetric=True, links=links.to(DEV), zeros=torch.zeros(64, device=DEV))
    close(got_l, want_l, rtol=2e-5, atol=2e-5 * float(want_s.abs().max()))


def test_grouped_slab_sum_vector_path_bit_identical(ops):
    """The grouped slab sum takes a 16-byte path for 4-lane sums of aligned operands (one thread = 4 elements
    x all four slab lanes): same order of additions as the dword kernel, for every slab count and epilogue."""
    g = torch.Generator().manual_seed(8)
    for n_slabs in (1, 2, 3, 4, 6, 8, 9, 17, 31, 40):
        for rows, cols in ((645, 32), (2048, 4), (1030, 12)):
            part = torch.randn(n_slabs, rows, cols, generator=g).to(DEV)
            scale = torch.rand(rows, generator=g).to(DEV)
            add = torch.randn(rows, cols, generator=g).to(DEV)
            gate = torch.randn(rows, cols, generator=g).to(DEV)
            base = torch.randn(rows, cols, generator=g).to(DEV)
            jobs = [ops.slab_job(part), ops.slab_job(part, alpha=0.25, row_scale=scale, addend=add, relu=True),
                    ops.slab_job(part, out=base.clone(), accumulate=True, gate=gate)]
            ops.gemm_group([], jobs)
            assert torch.equal(jobs[0].out, ops.sum_slabs(part))
            assert torch.equal(jobs[1].out, ops.sum_slabs(part, alpha=0.25, row_scale=scale, addend=add, relu=True))
            want = ops.sum_slabs(part, out=base.clone(), accumulate=True)
            assert torch.equal(jobs[2].out, torch.where(gate > 0, want, torch.zeros_like(want)))
            torch.testing.assert_close(jobs[0].out.cpu(), part.double().sum(0).float().cpu(), rtol=1e-5, atol=1e-5)


# ------------------------------------------------------------------ row-wise glue
def test_rowwise_ops(ops):
    g = torch.Generator().manual_seed(2)
    x = torch.randn(1000, 48, generator=g)
    xd = x.to(DEV)
    assert torch.equal(ops.transpose(xd).cpu(), x.t().contiguous())
    mul, div = torch.rand(1000, generator=g) + 0.5, torch.rand(1000, generator=g) + 0.5
    gate = torch.randn(1000, 48, generator=g)
    want = x * mul.unsqueeze(1) / div.unsqueeze(1) * (gate > 0)
    close(ops.rows_affine(xd, row_mul=mul.to(DEV), row_div=div.to(DEV), gate=gate.to(DEV)), want, rtol=1e-6)
    out = torch.ones(1000, 64, device=DEV)
    ops.rows_affine(xd, out=out[:, 8:56], accumulate=True)
    close(out[:, 8:56], x.double() + 1, rtol=1e-6)
    close(ops.col_sum(xd), x.double().sum(0), rtol=1e-5)
    big = torch.randn(19081, 32, generator=g)
    close(ops.col_sum(big.to(DEV)), big.double().sum(0), rtol=1e-5, atol=1e-3)
    wide = torch.randn(300, 700, generator=g)
    close(ops.col_sum(wide.to(DEV)), wide.double().sum(0), rtol=1e-5, atol=1e-3)


# ------------------------------------------------------------------ decoder
@pytest.mark.parametrize('k,idx_dtype', [(16, torch.int64), (16, torch.int32), (6, torch.int64), (4, torch.int32)])
def test_distmult_fwd_bwd(ops, k, idx_dtype):
    g = torch.Generator().manual_seed(k)
    n, r, m = 83, 9, 5000
    z = torch.randn(n, k, generator=g)
    w = torch.randn(r, k, generator=g) * 0.5
    sizes = torch.tensor([1, 900, 50, 0, 2049, 700, 300, 1000 - 1, 1])
    et = torch.repeat_interleave(torch.arange(r), sizes)
    idx = torch.randint(0, n, (2, m), generator=g)
    up = torch.randn(m, generator=g)
    zd, wd = z.to(DEV), w.to(DEV)
    idd, etd = idx.to(DEV, idx_dtype), et.to(DEV, idx_dtype)
    for sig in (True, False):
        s = ops.distmult_fwd(zd, wd, idd, etd, sigmoid=sig)
        close(s, O.distmult_fwd(z.double(), idx, et, w.double(), sig))
        gz, gw = ops.distmult_bwd(up.to(DEV), s, zd, wd, idd, etd, sigmoid=sig)
        wz, ww = O.distmult_bwd(up.double(), z.double(), idx, et, w.double(), sig)
        close(gz, wz, atol=1e-4)
        close(gw, ww, atol=1e-4)
    # shuffled relation order (non-uniform waves) takes the per-lane atomic path
    perm = torch.randperm(m, generator=g)
    gz2, gw2 = ops.distmult_bwd(up[perm].to(DEV), None, zd, wd, idd[:, perm.to(DEV)], etd[perm.to(DEV)], sigmoid=False)
    wz, ww = O.distmult_bwd(up.double(), z.double(), idx, et, w.double(), False)
    close(gz2, wz, atol=1e-4)
    close(gw2, ww, atol=1e-4)


def test_distmult_fused_objective(ops):
    g = torch.Generator().manual_seed(3)
    n, r, k, m = 645, 40, 16, 60000
    z = torch.randn(n, k, generator=g) * 0.7
    w = torch.randn(r, k, generator=g) * 0.5
    et = torch.sort(torch.randint(0, r, (m,), generator=g)).values
    pos = torch.randint(0, n, (2, m), generator=g)
    neg = torch.randint(0, n, (2, m), generator=g)
    zd, wd = z.to(DEV), w.to(DEV)
    loss, gz, gw = ops.distmult_loss(zd, wd, pos.to(DEV), neg.to(DEV), et.to(DEV))
    z64, w64 = z.double(), w.double()
    ps, ns = O.distmult_fwd(z64, pos, et, w64), O.distmult_fwd(z64, neg, et, w64)
    close(loss, O.tip_loss(ps, ns).view(1), rtol=2e-5)
    gp, gn = O.tip_loss_bwd(ps, ns)
    gz1, gw1 = O.distmult_bwd(gp, z64, pos, et, w64)
    gz2, gw2 = O.distmult_bwd(gn, z64, neg, et, w64)
    close(gz, gz1 + gz2, atol=2e-6)
    close(gw, gw1 + gw2, atol=2e-6)
    loss_only, a, b = ops.distmult_loss(zd, wd, pos.to(DEV), neg.to(DEV), et.to(DEV), need_grad=False)
    assert a is None and b is None
    close(loss_only, O.tip_loss(ps, ns).view(1), rtol=2e-5)
    # narrower decoders (distmult_objective_kernel<., 8 | 4>: 8 / 16 positions per scatter instruction, shuffles instead of
    # DPP row broadcasts), int32 ids, and the k/4-lanes-per-position task kernel as cross-check
    from tip_amd import _lib
    for kk, idt in ((8, torch.int64), (4, torch.int32), (16, torch.int32)):
        zk, wk = z[:, :kk].contiguous(), w[:, :kk].contiguous()
        args = (zk.to(DEV), wk.to(DEV), pos.to(DEV, idt), neg.to(DEV, idt), et.to(DEV, idt))
        lk, gzk, gwk = ops.distmult_loss(*args)
        psk, nsk = O.distmult_fwd(zk.double(), pos, et, wk.double()), O.distmult_fwd(zk.double(), neg, et, wk.double())
        close(lk, O.tip_loss(psk, nsk).view(1), rtol=2e-5)
        gpk, gnk = O.tip_loss_bwd(psk, nsk)
        a1, b1 = O.distmult_bwd(gpk, zk.double(), pos, et, wk.double())
        a2, b2 = O.distmult_bwd(gnk, zk.double(), neg, et, wk.double())
        close(gzk, a1 + a2, atol=2e-6)
        close(gwk, b1 + b2, atol=2e-6)
        again = ops.distmult_loss(*args)
        assert torch.equal(again[0], lk) and torch.equal(again[1], gzk) and torch.equal(again[2], gwk)
        _lib.set_option('dm_task_kernel', 1)
        lt, gzt, gwt = ops.distmult_loss(*args)
        _lib.set_option('dm_task_kernel', 0)
        close(lt, lk.cpu(), rtol=1e-5)
        close(gzt, gzk.cpu(), rtol=1e-4, atol=1e-7)
        close(gwt, gwk.cpu(), rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize('k', [16, 8, 4])
def test_distmult_fused_objective_packed_pairs_bit_identical(ops, k):
    """idx_bytes = 2 (include/tipk.h section 4): the pairs as one 32-bit word u | v << 16.  Same kernel, same order of
    operations -> loss, d z and d w are BIT-IDENTICAL to the int64-id call; the sampler's packed output holds exactly the
    pairs of its int64 output; shapes the packed kernel does not take fall back to plain ids with the same result."""
    from tip_amd import neg_sampling as NS
    g = torch.Generator().manual_seed(21)
    n, r = 645, 37
    sizes = torch.randint(0, 3000, (r,), generator=g)
    halves = [torch.randint(0, n, (2, int(c)), generator=g) for c in sizes]
    pos = torch.cat([torch.cat([h, h.flip(0)], dim=1) for h in halves], dim=1).to(DEV)      # mirrored like TIP's positives
    et = torch.repeat_interleave(torch.arange(r), 2 * sizes).to(DEV)
    ptr = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(2 * sizes, 0)])
    rg = torch.stack([ptr[:-1], ptr[1:]], 1)
    neg = NS.typed_negative_sampling(pos, n, rg, seed=77)
    neg_p = NS.typed_negative_sampling(pos, n, rg, seed=77, packed=True)
    assert neg_p.dtype == torch.int32 and neg_p.shape == (pos.shape[1],) and neg_p._tipk_packed_pairs
    assert torch.equal(ops.unpack_pairs(neg_p), neg)
    assert torch.equal(ops.unpack_pairs(ops.packed_pairs(pos, n)), pos)
    z = (torch.randn(n, k, generator=g) * 0.7).to(DEV)
    w = (torch.randn(r, k, generator=g) * 0.5).to(DEV)
    want = ops.distmult_loss(z, w, pos, neg, et)
    got = ops.distmult_loss(z, w, pos, neg_p, et)
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    lo = ops.distmult_loss(z, w, pos, neg_p, et, need_grad=False)
    assert torch.equal(lo[0], want[0]) and lo[1] is None
    # ids up to 65534 survive the 16-bit halves (sign bit of the word included)
    n_big = 65535
    pos_b = torch.randint(0, n_big, (2, 5000), generator=g).to(DEV)
    pos_b[:, :4] = torch.tensor([[65534, 0, 65534, 32768], [65534, 65534, 0, 32767]], device=DEV)
    assert torch.equal(ops.unpack_pairs(ops.packed_pairs(pos_b, n_big)), pos_b)
    et_b = torch.sort(torch.randint(0, 5, (5000,), generator=g)).values.to(DEV)
    cnt = torch.bincount(et_b.cpu(), minlength=5)
    pb = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(cnt, 0)])
    rg_b = torch.stack([pb[:-1], pb[1:]], 1)
    nb = NS.typed_negative_sampling(pos_b, n_big, rg_b, seed=5)
    nbp = NS.typed_negative_sampling(pos_b, n_big, rg_b, seed=5, packed=True)
    assert torch.equal(ops.unpack_pairs(nbp), nb)
    zb = (torch.randn(n_big, k, generator=g) * 0.5).to(DEV)
    wb = (torch.randn(5, k, generator=g) * 0.5).to(DEV)
    # (z too large for the LDS image: both calls run the generic float-atomic kernel on plain ids)
    for a, b in zip(ops.distmult_loss(zb, wb, pos_b, nbp, et_b), ops.distmult_loss(zb, wb, pos_b, nb, et_b)):
        close(a, b.cpu(), rtol=1e-4, atol=1e-7)
    # a width the objective kernel does not take (k = 32): packed negatives are widened, result as with plain ids
    z32 = (torch.randn(n, 32, generator=g) * 0.5).to(DEV)
    w32 = (torch.randn(r, 32, generator=g) * 0.5).to(DEV)
    for a, b in zip(ops.distmult_loss(z32, w32, pos, neg_p, et), ops.distmult_loss(z32, w32, pos, neg, et)):
        close(a, b.cpu(), rtol=1e-4, atol=1e-7)


def test_distmult_fused_objective_mirrored_positives(ops, monkeypatch):
    """TIP's positives list every pair in both directions per relation: the mirrored half is skipped and
    the first half counted twice -- same loss and gradients as the plain evaluation and as the oracle."""
    g = torch.Generator().manual_seed(8)
    n, r, k = 645, 23, 16
    sizes = torch.randint(1, 4000, (r,), generator=g)
    sizes[3] = 0
    halves = [torch.randint(0, n, (2, int(c)), generator=g) for c in sizes]
    pos = torch.cat([torch.cat([h, h.flip(0)], dim=1) for h in halves], dim=1)
    et = torch.repeat_interleave(torch.arange(r), 2 * sizes)
    m = pos.shape[1]
    neg = torch.randint(0, n, (2, m), generator=g)
    z = torch.randn(n, k, generator=g) * 0.7
    w = torch.randn(r, k, generator=g) * 0.5
    zd, wd, posd, negd, etd = z.to(DEV), w.to(DEV), pos.to(DEV), neg.to(DEV), et.to(DEV)
    tasks = ops.relation_tasks(etd, posd)
    assert set(tasks[:, 3].tolist()) == {0, 2}
    loss, gz, gw = ops.distmult_loss(zd, wd, posd, negd, etd)
    z64, w64 = z.double(), w.double()
    ps, ns = O.distmult_fwd(z64, pos, et, w64), O.distmult_fwd(z64, neg, et, w64)
    close(loss, O.tip_loss(ps, ns).view(1), rtol=2e-5)
    gp, gn = O.tip_loss_bwd(ps, ns)
    gz1, gw1 = O.distmult_bwd(gp, z64, pos, et, w64)
    gz2, gw2 = O.distmult_bwd(gn, z64, neg, et, w64)
    close(gz, gz1 + gz2, atol=2e-6)
    close(gw, gw1 + gw2, atol=2e-6)
    monkeypatch.setenv('TIPK_NO_SYMMETRIC_POS', '1')
    posd2 = posd.clone()                                               # new tensor: new cache entry
    assert bool((ops.relation_tasks(etd, posd2)[:, 3] == 1).all())
    loss2, gz2d, gw2d = ops.distmult_loss(zd, wd, posd2, negd, etd)
    close(loss2, loss.cpu(), rtol=1e-5)
    close(gz2d, gz.cpu(), rtol=1e-4, atol=1e-6)
    close(gw2d, gw.cpu(), rtol=1e-4, atol=1e-6)


def test_distmult_fused_objective_is_bitwise_reproducible(ops, monkeypatch):
    """The objective's sums across workgroups go through 64-bit fixed-point integer atomics (exact,
    order-independent): loss, d z and d w are identical bit for bit from call to call at full BioSNAP size
    (8.3 M triples, 256 workgroups), and agree with the float-atomic path to rounding."""
    from tip_amd.data import build_data_dict
    from tip_amd.neg_sampling import typed_negative_sampling
    dd = build_data_dict()
    pos, et, rg = dd['dd_train_idx'].to(DEV), dd['dd_train_et'].to(DEV), dd['dd_train_range']
    neg = typed_negative_sampling(pos, dd['n_drug'], rg, seed=3)
    g = torch.Generator().manual_seed(0)
    z = (torch.randn(dd['n_drug'], 16, generator=g) * 0.5).to(DEV)
    w = (torch.randn(dd['n_dd_et'], 16, generator=g) * 0.25).to(DEV)
    runs = [ops.distmult_loss(z, w, pos, neg, et) for _ in range(4)]
    for loss, gz, gw in runs[1:]:
        assert torch.equal(loss, runs[0][0]) and torch.equal(gz, runs[0][1]) and torch.equal(gw, runs[0][2])
    assert float(runs[0][1].abs().max()) > 0 and float(runs[0][2].abs().max()) > 0
    monkeypatch.setenv('TIPK_FLOAT_ATOMICS', '1')
    loss_f, gz_f, gw_f = ops.distmult_loss(z, w, pos, neg, et)
    close(runs[0][0], loss_f.cpu(), rtol=1e-5)
    close(runs[0][1], gz_f.cpu(), rtol=1e-4, atol=1e-7)
    close(runs[0][2], gw_f.cpu(), rtol=1e-4, atol=1e-8)


# ------------------------------------------------------------------ negative sampler
def test_negative_sampler_bit_exact_vs_spec_and_properties():
    from tip_amd import neg_sampling as NS
    rng = np.random.RandomState(4)
    n = 41
    sizes = [0, 300, 7, 1200, 1]                     # 1200 of 1681 cells -> dense relation, many rejections
    pos = np.concatenate([rng.randint(0, n, (2, s)) for s in sizes], axis=1).astype(np.int64)
    rel_ptr = np.r_[0, np.cumsum(sizes)]
    rg = torch.tensor(np.stack([rel_ptr[:-1], rel_ptr[1:]], 1))
    pos_t = torch.from_numpy(pos).to(DEV)
    got = NS.typed_negative_sampling(pos_t, n, rg, seed=0x1234567887654321)
    assert got.dtype == torch.int64 and got.shape == pos_t.shape and got.device == pos_t.device
    want = typed_negative_sampling_spec(pos, n, rel_ptr, 0x1234567887654321)
    assert np.array_equal(got.cpu().numpy(), want)                  # LDS-bitmap kernel (n^2 bits fit)
    import os
    os.environ['TIPK_NO_BITMAP'] = '1'                              # binary-search kernel: same bits
    try:
        got_bs = NS.typed_negative_sampling(pos_t, n, rg, seed=0x1234567887654321)
    finally:
        del os.environ['TIPK_NO_BITMAP']
    assert torch.equal(got, got_bs)
    gk = (got[0] * n + got[1]).cpu().numpy()
    assert gk.min() >= 0 and gk.max() < n * n
    for r in range(len(sizes)):                      # no sampled pair is a positive of its relation
        a, b = rel_ptr[r], rel_ptr[r + 1]
        assert not np.isin(gk[a:b], pos[0, a:b] * n + pos[1, a:b]).any()
    # stream semantics: manual_seed reproduces, consecutive calls differ; call n uses the key
    # call_key(seed, n) derived ON THE DEVICE from the device-resident call counter
    from oracle.philox_sampler import call_key
    NS.manual_seed(7)
    a1, a2 = NS.typed_negative_sampling(pos_t, n, rg), NS.typed_negative_sampling(pos_t, n, rg)
    NS.manual_seed(7)
    b1 = NS.typed_negative_sampling(pos_t, n, rg)
    assert torch.equal(a1, b1) and not torch.equal(a1, a2)
    assert np.array_equal(a1.cpu().numpy(), typed_negative_sampling_spec(pos, n, rel_ptr, call_key(7, 0)))
    assert np.array_equal(a2.cpu().numpy(), typed_negative_sampling_spec(pos, n, rel_ptr, call_key(7, 1)))
    # relation-sharded runs: a rank that holds relations (3, 1) and passes their global offsets draws exactly the
    # unsharded run's negatives for them (Philox counters run over GLOBAL positions), on both kernels and in the spec
    keep = [1, 3]
    loc = np.concatenate([pos[:, rel_ptr[r]:rel_ptr[r + 1]] for r in keep], axis=1)
    loc_ptr = np.r_[0, np.cumsum([sizes[r] for r in keep])]
    off = torch.tensor([rel_ptr[r] - loc_ptr[i] for i, r in enumerate(keep)])
    rg_l = torch.tensor(np.stack([loc_ptr[:-1], loc_ptr[1:]], 1))
    want_l = np.concatenate([want[:, rel_ptr[r]:rel_ptr[r + 1]] for r in keep], axis=1)
    got_l = NS.typed_negative_sampling(torch.from_numpy(loc).to(DEV), n, rg_l, seed=0x1234567887654321, pos_offset=off)
    assert np.array_equal(got_l.cpu().numpy(), want_l)
    assert np.array_equal(typed_negative_sampling_spec(loc, n, loc_ptr, 0x1234567887654321, pos_offset=off.numpy()), want_l)
    assert call_key(7, 1) == NS.call_key(7, 1)
    # the stream's seed lives on the device next to its position: a captured sampler call keeps
    # advancing on replay, and RE-SEEDING AFTER CAPTURE takes effect in the replays
    NS.manual_seed(7)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            captured = NS.typed_negative_sampling(pos_t, n, rg)
    torch.cuda.synchronize()
    assert NS.stream_position(DEV) == 0                               # capture executes nothing
    graph.replay(); torch.cuda.synchronize()
    assert torch.equal(captured, a1) and NS.stream_position(DEV) == 1
    graph.replay(); torch.cuda.synchronize()
    assert torch.equal(captured, a2)
    NS.manual_seed(99)                                                # other seed, after capture
    graph.replay(); torch.cuda.synchronize()
    assert np.array_equal(captured.cpu().numpy(), typed_negative_sampling_spec(pos, n, rel_ptr, call_key(99, 0)))
    NS.manual_seed(7)
    graph.replay(); torch.cuda.synchronize()
    assert torch.equal(captured, a1)


def test_negative_sampler_uniformity_vs_reference_distribution():
    """Same distribution as the reference's sampler: uniform over the non-positive cells of the
    relation (chi-square against the expected flat histogram; the reference restatement under a
    seeded numpy generator is held to the same bar)."""
    from tip_amd import neg_sampling as NS
    rng = np.random.RandomState(8)
    n = 16
    pos = rng.randint(0, n, (2, 60)).astype(np.int64)
    pos_big = np.tile(pos, (1, 2000))                                  # 120 000 draws, same positives
    rg = torch.tensor([[0, pos_big.shape[1]]])
    got = NS.typed_negative_sampling(torch.from_numpy(pos_big).to(DEV), n, rg, seed=99).cpu().numpy()
    keys = got[0] * n + got[1]
    free = np.setdiff1d(np.arange(n * n), pos[0] * n + pos[1])
    hist = np.bincount(keys, minlength=n * n)[free]
    exp = keys.size / free.size
    chi2 = float(((hist - exp) ** 2 / exp).sum())
    assert chi2 < free.size + 6 * np.sqrt(2 * free.size), chi2        # ~ mean + 6 sigma
    ref = O.typed_negative_sampling(torch.from_numpy(pos_big), n, rg, np.random.RandomState(1)).numpy()
    rk = ref[0] * n + ref[1]
    rhist = np.bincount(rk, minlength=n * n)[free]
    # the reference leaks a few positives (its resample quirk); on the free cells it is flat too
    chi2_ref = float(((rhist - rhist.sum() / free.size) ** 2 / (rhist.sum() / free.size)).sum())
    assert chi2_ref < free.size + 6 * np.sqrt(2 * free.size)


# ------------------------------------------------------------------ relation-local (LDS) gather
@pytest.mark.parametrize('d', [4, 16, 32, 128])
def test_rel_gather_fwd_bwd(ops, d):
    from tip_amd.plan import build_rel_plan
    from tip_amd import _lib
    g = torch.Generator().manual_seed(d)
    N, R = 645, 9
    sizes = torch.tensor([40000, 1, 0, 3000, 17000, 250, 5, 9000, 700])          # > 16384: several id chunks
    E = int(sizes.sum())
    rel = torch.repeat_interleave(torch.arange(R), sizes)
    src = torch.randint(0, N, (E,), generator=g)
    dst = torch.randint(0, N - 7, (E,), generator=g)
    dst[:9000] = 11                                                            # hub (longest run 9000 edges)
    assert _lib.lib().tipk_rel_gather_supported(N, d, 0) >= 1 and _lib.lib().tipk_rel_gather_supported(N, d, 1) >= 1
    y = torch.randn(R * N, d, generator=g)
    plan = build_rel_plan(dst, src, rel, N, R, n_wg=256).to(DEV)
    got = ops.rel_gather(plan, y.to(DEV), backward=False)
    close(got, O.gather_sum(y.double(), rel * N + src, dst, N))
    assert torch.equal(got, ops.rel_gather(plan, y.to(DEV), backward=False))    # bitwise reproducible
    gp = torch.randn(N, d, generator=g)
    planb = build_rel_plan(src, dst, rel, N, R, n_wg=256, backward=True).to(DEV)
    gotb = ops.rel_gather(planb, gp.to(DEV), backward=True)
    close(gotb, O.gather_sum(gp.double(), dst, rel * N + src, R * N))
    # big relations dealt to several work units (the BioSNAP plans do this for the top relations)
    plan_u = build_rel_plan(dst, src, rel, N, R, n_wg=256, max_unit=5000).to(DEV)
    assert plan_u.n_units > R
    close(ops.rel_gather(plan_u, y.to(DEV), backward=False), O.gather_sum(y.double(), rel * N + src, dst, N))
    planb_u = build_rel_plan(src, dst, rel, N, R, n_wg=256, backward=True, max_unit=5000).to(DEV)
    out_u = ops.rel_gather(planb_u, gp.to(DEV), backward=True)
    close(out_u, O.gather_sum(gp.double(), dst, rel * N + src, R * N))
    assert torch.equal(out_u, ops.rel_gather(planb_u, gp.to(DEV), backward=True))
    # few workgroups (several relations each) and strided table
    plan3 = build_rel_plan(dst, src, rel, N, R, n_wg=2).to(DEV)
    wide = torch.randn(R * N, d + 8, generator=g).to(DEV)
    close(ops.rel_gather(plan3, wide[:, 4:4 + d], backward=False),
          O.gather_sum(wide[:, 4:4 + d].cpu().double(), rel * N + src, dst, N))


@pytest.mark.parametrize('N,d', [(1000, 32), (1000, 16), (1024, 8), (300, 64), (17, 128)])
def test_rel_gather_other_node_counts(ops, N, d):
    """Node counts other than BioSNAP's 645: the table-prefetch width (TU) and the column split change."""
    from tip_amd.plan import build_rel_plan
    from tip_amd import _lib
    g = torch.Generator().manual_seed(N + d)
    R = 5
    sizes = torch.tensor([30000, 0, 7, 12000, 2500])
    E = int(sizes.sum())
    rel = torch.repeat_interleave(torch.arange(R), sizes)
    src = torch.randint(0, N, (E,), generator=g)
    dst = torch.randint(0, max(1, N - 3), (E,), generator=g)
    split_f = _lib.lib().tipk_rel_gather_supported(N, d, 0)
    split_b = _lib.lib().tipk_rel_gather_supported(N, d, 1)
    assert split_f >= 1 and split_b >= 1
    y = torch.randn(R * N, d, generator=g)
    plan = build_rel_plan(dst, src, rel, N, R, n_wg=max(1, 256 // split_f)).to(DEV)
    close(ops.rel_gather(plan, y.to(DEV), backward=False), O.gather_sum(y.double(), rel * N + src, dst, N))
    gp = torch.randn(N, d, generator=g)
    scale = torch.rand(N, generator=g) + 0.5
    planb = build_rel_plan(src, dst, rel, N, R, n_wg=256, backward=True).to(DEV)
    close(ops.rel_gather(planb, gp.to(DEV), backward=True, row_scale=scale.to(DEV)),
          O.gather_sum(gp.double() * scale.double().unsqueeze(1), dst, rel * N + src, R * N))


@pytest.mark.parametrize('N,d', [(645, 32), (645, 16), (645, 4), (645, 128), (1000, 32), (1024, 8), (300, 64), (17, 128), (2000, 16)])
def test_rel_stream_bwd(ops, N, d):
    """Wave-stream transposed pass (include/tipk.h section 1d): every (relation, source) row, hub runs that
    span many bands of one slot, relations without edges, strided table, row scale, few workgroups."""
    from tip_amd.plan import build_stream_plan
    g = torch.Generator().manual_seed(N + d)
    R = 9
    sizes = torch.tensor([40000, 1, 0, 3000, 17000, 250, 5, 9000, 700])
    E = int(sizes.sum())
    rel = torch.repeat_interleave(torch.arange(R), sizes)
    src = torch.randint(0, N, (E,), generator=g)
    dst = torch.randint(0, max(1, N - 7), (E,), generator=g)
    src[:9000] = 11                                                            # hub: one output row with 9 000 edges
    split = ops.rel_stream_split(N, d)
    assert split >= 1
    lanes = d // split // 4
    gp = torch.randn(N, d, generator=g)
    scale = torch.rand(N, generator=g) + 0.5
    want = O.gather_sum(gp.double() * scale.double().unsqueeze(1), dst, rel * N + src, R * N)
    for n_wg in (256, 3):
        sp = build_stream_plan(src, dst, rel, N, R, n_wg, lanes, ops.rel_stream_piece()).to(DEV)
        got = ops.rel_stream_bwd(sp, gp.to(DEV), row_scale=scale.to(DEV))
        close(got, want)
        assert torch.equal(got, ops.rel_stream_bwd(sp, gp.to(DEV), row_scale=scale.to(DEV)))     # bitwise reproducible
    wide = torch.randn(N, d + 8, generator=g).to(DEV)
    close(ops.rel_stream_bwd(sp, wide[:, 4:4 + d]), O.gather_sum(wide[:, 4:4 + d].cpu().double(), dst, rel * N + src, R * N))
    from tip_amd import _lib
    assert _lib.lib().tipk_stream_gather_supported(10000, 32, 4) == 0 and _lib.lib().tipk_stream_gather_supported(645, 24, 4) == 0
    assert _lib.lib().tipk_stream_gather_supported(19081, 32, 16) == 16 and _lib.lib().tipk_stream_gather_supported(19081, 16, 16) == 8
    assert _lib.lib().tipk_stream_gather_supported(19081, 32, 4) == 0 and _lib.lib().tipk_stream_gather_supported(30000, 32, 16) == 0


@pytest.mark.parametrize('N,R,nb', [(645, 1097, 32), (200, 50, 32), (100, 1500, 16), (64, 9, 8)])
def test_stream_gather_pair_form(ops, N, R, nb):
    """The forward pass in pair form: cell (v, u) = sum of att[r] over the relations linking u -> v, written into a
    buffer that was zeroed once (cells of unlinked pairs are never touched); table = att stays in LDS."""
    from tip_amd.plan import build_stream_plan_rows
    g = torch.Generator().manual_seed(N + R)
    E = 60000
    rel = torch.randint(0, R, (E,), generator=g)
    src = torch.randint(0, N, (E,), generator=g)
    dst = (src + 1 + torch.randint(0, max(1, N // 4), (E,), generator=g)) % N      # pairs repeat: ~E / (N * N / 4) relations each
    split = ops.stream_gather_split(R, nb)
    assert split >= 1
    sp = build_stream_plan_rows(dst * N + src, rel, N * N, R, 64, nb // split // 4, ops.rel_stream_piece()).to(DEV)
    att = torch.randn(R, nb, generator=g)
    cells = torch.zeros(N * N, nb, device=DEV)
    for _ in range(2):                                                             # second call rewrites the same cells
        ops.stream_gather(sp, att.to(DEV), write_zeros=False, out=cells, kind=1)
    want = torch.zeros(N * N, nb, dtype=torch.float64).index_add_(0, dst * N + src, att.double()[rel])
    close(cells, want)
    assert int((cells.abs().sum(1) > 0).sum()) == int(torch.unique(dst * N + src).numel())


def test_rel_gather_unsupported_shapes():
    from tip_amd import _lib
    L = _lib.lib()
    assert L.tipk_rel_gather_supported(645, 24, 0) == 0     # not a power of two
    assert L.tipk_rel_gather_supported(10000, 32, 0) == 0 and L.tipk_rel_gather_supported(10000, 32, 1) == 0
    assert L.tipk_rel_gather_supported(2000, 16, 1) == 0    # node tables are prefetched by 1024 threads
    for d in (16, 32, 64, 128):                             # wide rows run as several column blocks
        assert L.tipk_rel_gather_supported(645, d, 0) >= 1 and L.tipk_rel_gather_supported(645, d, 1) >= 1
    assert L.tipk_rel_gather_supported(645, 32, 0) == 2 and L.tipk_rel_gather_supported(645, 32, 1) == 1   # column blocks


# ------------------------------------------------------------------ ranking metrics on device
def test_rank_metrics_vs_sklearn():
    from tip_amd import ops as O2
    from tip_amd.utils import auprc_auroc_ap_by_range
    rng = np.random.RandomState(2)
    sizes = [1, 3, 700, 40, 8192, 5000, 2]
    ptr = np.r_[0, np.cumsum(sizes)]
    tot = int(ptr[-1])
    pos = (rng.rand(tot) * 0.6 + 0.3).astype(np.float32)
    neg = (rng.rand(tot) * 0.7).astype(np.float32)
    a, b = ptr[2], ptr[3]                              # heavy ties in one relation
    pos[a:b] = np.round(pos[a:b], 1)
    neg[a:b] = np.round(neg[a:b], 1)
    neg[ptr[3]:ptr[4]] = pos[ptr[3]:ptr[4]]            # identical score lists: AUROC 0.5 by ties
    rg = torch.tensor(np.stack([ptr[:-1], ptr[1:]], 1))
    got = auprc_auroc_ap_by_range(torch.from_numpy(pos).to(DEV), torch.from_numpy(neg).to(DEV), rg)
    assert got.shape == (3, len(sizes))
    for r in range(len(sizes)):
        y = np.r_[np.ones(sizes[r]), np.zeros(sizes[r])]
        s = np.r_[pos[ptr[r]:ptr[r + 1]], neg[ptr[r]:ptr[r + 1]]]
        np.testing.assert_allclose(got[:, r], O.auprc_auroc_ap(y, s), rtol=1e-9, atol=1e-12)
    assert abs(got[1, 3] - 0.5) < 1e-12
    # a relation beyond the single-workgroup sort falls back to the host path with the same numbers
    big = 9000
    p2, n2 = rng.rand(big).astype(np.float32), rng.rand(big).astype(np.float32)
    assert O2.rank_metrics(torch.from_numpy(p2).to(DEV), torch.from_numpy(n2).to(DEV),
                           torch.tensor([0, big], device=DEV), big) is None
    got2 = auprc_auroc_ap_by_range(torch.from_numpy(p2).to(DEV), torch.from_numpy(n2).to(DEV), torch.tensor([[0, big]]))
    np.testing.assert_allclose(got2[:, 0], O.auprc_auroc_ap(np.r_[np.ones(big), np.zeros(big)], np.r_[p2, n2]), rtol=1e-9)


# ------------------------------------------------------------------ device-side train/test split
def test_device_split_bit_exact_vs_spec():
    """tipk_split_flags + tipk_split_scatter == oracle/philox_split.py (the reference's post-draw layout:
    kept pairs in list order, mirrored halves appended, edge types, ranges) -- empty / single-pair / large
    relations, int64 and uint16 pair ids, p in {0, 0.9, 1}."""
    from oracle.philox_split import process_edges_spec
    from tip_amd.data import device_process_edges
    rng = np.random.RandomState(2)
    sizes = [0, 700, 1, 0, 4097, 256, 255, 257, 30000]
    ptr = np.r_[0, np.cumsum(sizes)].astype(np.int64)
    pairs = np.stack([rng.randint(0, 645, ptr[-1]), rng.randint(0, 645, ptr[-1])]).astype(np.int64)
    for p, seed in ((0.9, 1111), (0.5, (1 << 63) + 12345), (1.0, 1), (0.0, 2)):
        want = process_edges_spec(pairs, ptr, p, seed)
        for dt in (torch.int64, torch.uint16):
            got = device_process_edges(torch.from_numpy(pairs).to(dt).to(DEV), torch.from_numpy(ptr).to(DEV), p, seed)
            for g_, w_ in zip(got, want):
                assert g_.dtype == torch.int64 and g_.is_cuda
                assert np.array_equal(g_.cpu().numpy(), w_), (p, dt)


def test_device_built_biosnap_data_dict_trains():
    """The whole ingest on device (blob -> HBM -> Philox split + mirroring of the 4.6 M D-D pairs and the
    P-P graph): schema of SURVEY 8(a) A0, the data contract (every block = [pairs | mirrored pairs]),
    Bernoulli(0.9) sizes, bit-exact against the host spec at full size, and a TIP model trains on it."""
    from oracle.philox_split import split_flags_spec
    from tip_amd.data import build_data_dict_device, BIOSNAP_BLOB
    from tip_amd.layers import TIP, Setting
    d = build_data_dict_device(DEV)
    z = np.load(BIOSNAP_BLOB)
    P = z['dd_pairs'].shape[1]
    R = d['n_dd_et']
    assert R == 1097 and d['dd_train_idx'].is_cuda and d['dd_train_idx'].dtype == torch.int64
    n_tr, n_te = d['dd_train_idx'].shape[1], d['dd_test_idx'].shape[1]
    assert n_tr + n_te == 2 * P and abs(n_tr / 2 - 0.9 * P) < 5 * np.sqrt(P * 0.09)
    take = split_flags_spec(P, 0.9, 1111)
    ptr = z['dd_ptr'].astype(np.int64)
    kept = np.add.reduceat(take.astype(np.int64), ptr[:-1])
    rg = d['dd_train_range'].cpu().numpy()
    assert np.array_equal(rg[:, 1] - rg[:, 0], 2 * kept)
    for r in (0, 1, 500, R - 1):                                             # contents of a few relations
        a, b = ptr[r], ptr[r + 1]
        sel = z['dd_pairs'][:, a:b][:, take[a:b]].astype(np.int64)
        blk = d['dd_train_idx'][:, rg[r, 0]:rg[r, 1]].cpu().numpy()
        assert np.array_equal(blk, np.concatenate([sel, sel[::-1]], 1))
        assert bool((d['dd_train_et'][rg[r, 0]:rg[r, 1]] == r).all())
    h = (rg[:, 1] - rg[:, 0]) // 2                                           # mirrored halves everywhere
    first = torch.cat([torch.arange(a, a + k) for a, k in zip(rg[:, 0].tolist(), h.tolist())]).to(DEV)
    second = torch.cat([torch.arange(a + k, a + 2 * k) for a, k in zip(rg[:, 0].tolist(), h.tolist())]).to(DEV)
    assert torch.equal(d['dd_train_idx'][:, first], d['dd_train_idx'][:, second].flip(0))
    pp = d['pp_train_indices']
    assert pp.shape[1] % 2 == 0 and torch.equal(pp[:, :pp.shape[1] // 2], pp[:, pp.shape[1] // 2:].flip(0))
    torch.manual_seed(0)
    model = TIP(Setting(), torch.device(DEV), data=d)
    opt = torch.optim.Adam(model.parameters(), lr=0.01)
    losses = []
    for _ in range(3):
        opt.zero_grad()
        loss = model()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert abs(losses[0] - 2 * np.log(2)) < 0.05 and losses[-1] < losses[0]
    rec = model.test(print_output=False)
    assert rec.shape == (3, R) and np.isfinite(rec).all()


# ------------------------------------------------------------------ CSR rows (transposed pass of large graphs)
@pytest.mark.parametrize('d', [8, 12, 16, 32, 64, 128, 200, 256])
def test_gather_rows_csr_vs_reference(ops, d):
    """tipk_gather_rows_csr == per-row sums in fp64: empty rows (also whole empty tasks, leading / trailing),
    a heavy row, row counts that are not a multiple of the rows per slot; bitwise reproducible."""
    from tip_amd.plan import build_csr_plan, execute_csr_reference
    g = torch.Generator().manual_seed(d)
    n_out, n_tab, E = 1003, 77, 2600
    out_row = torch.randint(40, n_out - 30, (E,), generator=g)               # rows < 40 and the last 30 stay empty
    out_row[:400] = 500                                                      # a heavy row
    out_row[400:420] = torch.arange(600, 620)
    tab_row = torch.randint(0, n_tab, (E,), generator=g)
    table = torch.randn(n_tab, d, generator=g)
    plan = build_csr_plan(out_row, tab_row, n_out, n_tab)
    want = execute_csr_reference(plan, table.double())
    dev_plan = build_csr_plan(out_row.to(DEV), tab_row.to(DEV), n_out, n_tab)
    got = ops.gather_rows_csr(dev_plan, table.to(DEV))
    close(got, want, rtol=1e-4, atol=2e-4)                                   # a 400-term fp32 sum in the heavy row
    assert torch.equal(got, ops.gather_rows_csr(dev_plan, table.to(DEV)))
    # strided table view and an all-empty graph
    wide = torch.randn(n_tab, d + 8, generator=g).to(DEV)
    close(ops.gather_rows_csr(dev_plan, wide[:, 4:4 + d]), execute_csr_reference(plan, wide[:, 4:4 + d].cpu().double()),
          rtol=1e-4, atol=2e-4)
    empty = build_csr_plan(torch.zeros(0, dtype=torch.long, device=DEV), torch.zeros(0, dtype=torch.long, device=DEV), 13, n_tab)
    assert float(ops.gather_rows_csr(empty, table.to(DEV)).abs().max()) == 0.0


@pytest.mark.parametrize('N,d,bias,relu', [(19081, 32, True, True), (19081, 16, False, False), (3000, 8, True, False)])
def test_stream_gather_8_byte_rows_with_epilogue(ops, N, d, bias, relu):
    """tipk_stream_gather with max_split = 16: a table too tall for 16-byte rows runs as 2-column blocks of 8-byte rows
    (the P-P graph: 19 081 proteins); epilogue relu?(out_scale * sum + bias).  Against an fp64 index_add, and the plan
    against its CPU interpreter."""
    from tip_amd.plan import build_stream_plan_rows, execute_stream_plan_reference
    g = torch.Generator().manual_seed(N + d)
    E = 60000 if N > 5000 else 20000
    dst = torch.randint(0, N, (E,), generator=g)
    src = torch.randint(0, N, (E,), generator=g)
    dst[:3000] = 7                                                     # a hub row (wide run) ...
    dst = torch.where(dst == 11, torch.full_like(dst, 12), dst)        # ... and a row without edges
    split = ops.stream_gather_split(N, d, max_split=16)
    assert split and (d // split == 2 or N < 5000)
    dc = d // split
    lanes, row_bytes = max(1, dc // 4), dc * 4
    sp = build_stream_plan_rows(dst.to(DEV), src.to(DEV), N, N, max(1, 256 // split), lanes, row_bytes=row_bytes)
    x = torch.randn(N, d, generator=g)
    pre, post = torch.rand(N, generator=g) + 0.5, torch.rand(N, generator=g) + 0.5
    b = torch.randn(d, generator=g) if bias else None
    got = ops.stream_gather(sp, x.to(DEV), row_scale=pre.to(DEV), out_scale=post.to(DEV), bias=None if b is None else b.to(DEV),
                            relu=relu, max_split=16)
    want = torch.zeros(N, d, dtype=torch.float64).index_add_(0, dst, (x.double() * pre.double()[:, None])[src]) * post.double()[:, None]
    if b is not None:
        want = want + b.double()
    if relu:
        want = want.clamp(min=0)
    close(got, want, rtol=1e-5, atol=1e-5 * float(want.abs().max()))
    if b is not None:
        close(got[11], (b.double().clamp(min=0) if relu else b.double()), atol=1e-7)     # the empty row: epilogue of 0
    again = ops.stream_gather(sp, x.to(DEV), row_scale=pre.to(DEV), out_scale=post.to(DEV), bias=None if b is None else b.to(DEV),
                              relu=relu, max_split=16)
    assert torch.equal(got, again)
    if N <= 5000:
        ref = execute_stream_plan_reference(sp.to('cpu'), x * pre[:, None])
        close(ops.stream_gather(sp, x.to(DEV), row_scale=pre.to(DEV), max_split=16), ref.double(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('n,n_src,p,q,ne,cat', [(645, 3640, 16, 16, 48, True), (645, 3640, 16, 64, 64, False), (37, 19, 8, 4, 5, True),
                                                (5, 300, 6, 10, 10, False), (130, 64, 64, 64, 3, True)])
def test_drug_mix_gather_fused_launches(ops, n, n_src, p, q, ne, cat):
    """tipk_drug_mix_gather_fwd / tipk_drug_mix_bwd (tipk.h section 3): P -> D mean (MyHierarchyConv, src/layers.py:229-242) + dense map +
    /d_norm + cat | add (:532-539) in one launch each way == the fp64 formula, drugs without targets, repeated edges,
    non-unit d_norm, a strided upstream gradient; bitwise reproducible."""
    from tip_amd.layers import hier_graph
    g = torch.Generator().manual_seed(n + p + q)
    E = 6 * n + 3
    src = torch.randint(0, n_src, (E,), generator=g)
    dst = torch.randint(0, max(1, n - 2), (E,), generator=g)          # the last two targets have no edge
    src[:4] = src[4:8]; dst[:4] = dst[4:8]                            # repeated edges
    ei = torch.stack([src, dst + n_src]).to(DEV)
    graph = hier_graph(ei, n_src + n, n_src, table_rows=n_src, d=p)
    xd = torch.randn(n, ne, generator=g)
    h = torch.randn(n_src, p, generator=g)
    w = torch.randn(p, q, generator=g)
    dn = torch.rand(n, generator=g) + 0.5
    cnt = torch.bincount(dst, minlength=n).clamp(min=1).double()
    mean = torch.zeros(n, p, dtype=torch.float64).index_add_(0, dst, h.double()[src]) / cnt.unsqueeze(1)
    mapped = mean @ w.double()
    base = xd.double() / dn.double().unsqueeze(1)
    want = torch.cat([base, mapped], 1) if cat else base + mapped
    xd_d, h_d, w_d = (t.to(DEV).requires_grad_() for t in (xd, h, w))
    out = ops.drug_mix_gather(xd_d, h_d, w_d, dn.to(DEV), cat, graph)
    close(out, want, rtol=2e-5, atol=2e-5)
    gup = torch.randn(n, want.shape[1] + 3, generator=g)[:, 1:-2]      # a column-slice view: strided rows
    out.backward(gup.to(DEV))
    gd = gup.double()
    g_pd = gd[:, ne:] if cat else gd
    close(xd_d.grad, (gd[:, :ne] / dn.double().unsqueeze(1)), rtol=2e-5, atol=2e-6)
    close(w_d.grad, mean.t() @ g_pd, rtol=2e-5, atol=2e-5 * float((mean.t() @ g_pd).abs().max()))
    g_mean = (g_pd @ w.double().t()) / cnt.unsqueeze(1)
    want_h = torch.zeros(n_src, p, dtype=torch.float64).index_add_(0, src, g_mean[dst])
    close(h_d.grad, want_h, rtol=2e-5, atol=2e-5 * float(want_h.abs().max()))
    xd2, h2, w2 = (t.to(DEV).requires_grad_() for t in (xd, h, w))
    out2 = ops.drug_mix_gather(xd2, h2, w2, dn.to(DEV), cat, graph)
    out2.backward(gup.to(DEV))
    assert torch.equal(out2, out) and torch.equal(h2.grad, h_d.grad) and torch.equal(w2.grad, w_d.grad)


# ------------------------------------------------------------------ round 6: the fused launches of the encoder-level schedule
@pytest.mark.parametrize('n,n_src,p,q,ne,cat,nb,d_out', [(645, 3640, 16, 16, 48, True, 32, 32), (37, 50, 8, 8, 12, True, 5, 16),
                                                         (130, 64, 16, 64, 64, False, 32, 32), (33, 300, 6, 10, 10, True, 3, 16)])
def test_drug_mix_gather_xb_fwd(ops, n, n_src, p, q, ne, cat, nb, d_out):
    """tipk_drug_mix_gather_xb_fwd: the forward launch of the P -> D stage that also leaves XB = x0 basis (node-major, rows
    padded to 32 columns) and x0 root (src/layers.py:526-545): x0 / mean bit for bit what tipk_drug_mix_gather_fwd writes, the
    products == fp64, exact on integers (a k-ordered fma chain), a hub drug with its own workgroup, drugs without targets."""
    from tip_amd import encoder
    from tip_amd.layers import hier_graph
    g = torch.Generator().manual_seed(n + p + q + nb)
    E = 6 * n + 3
    src = torch.randint(0, n_src, (E,), generator=g)
    dst = torch.randint(0, max(1, n - 2), (E,), generator=g)
    if n >= 100:
        dst[:700] = 3                                                  # a hub: more than 512 targets -> all 16 wavefronts on it
        dst[700:900], dst[900:1000], dst[1000:1070] = 5, 9, 11         # 65 ... 512 targets: four wavefronts each
    ei = torch.stack([src, dst + n_src]).to(DEV)
    graph = hier_graph(ei, n_src + n, n_src, table_rows=n_src, d=p)
    cols = ne + q if cat else ne
    if not ops.lib().tipk_drug_mix_gather_xb_supported(p, q, ne, int(cat), nb, d_out):
        assert cols % 4 != 0 or d_out not in (16, 32)
        pytest.skip('shape not taken by the fused launch')
    xd, h, w = torch.randn(n, ne, generator=g), torch.randn(n_src, p, generator=g), torch.randn(p, q, generator=g)
    dn = torch.rand(n, generator=g) + 0.5
    basis, root = torch.randn(nb, cols, d_out, generator=g), torch.randn(cols, d_out, generator=g)
    n_pad = -(-n // 8) * 8

    def run(xd, h, w, basis, root):
        xb_pad = torch.zeros(n_pad, nb, 32, device=DEV)
        x0, mean, xroot = encoder.drug_mix_gather_xb(xd.to(DEV), h.to(DEV), w.to(DEV), dn.to(DEV), cat, graph, basis.to(DEV),
                                                     root.to(DEV), xb_pad[:, :, :d_out])
        return x0, mean, xroot, xb_pad
    x0, mean, xroot, xb_pad = run(xd, h, w, basis, root)
    ref = ops.drug_mix_gather(xd.to(DEV), h.to(DEV), w.to(DEV), dn.to(DEV), cat, graph)
    assert torch.equal(x0, ref)
    want_b = torch.einsum('nk,bkc->nbc', x0.double().cpu(), basis.double())
    want_r = x0.double().cpu() @ root.double()
    close(xb_pad[:n, :, :d_out], want_b, rtol=2e-5, atol=2e-5 * float(want_b.abs().max()))
    close(xroot, want_r, rtol=2e-5, atol=2e-5 * float(want_r.abs().max()))
    assert float(xb_pad[n:].abs().max()) == 0.0 and float(xb_pad[:, :, d_out:].abs().max() if d_out < 32 else 0.0) == 0.0
    again = run(xd, h, w, basis, root)
    assert all(torch.equal(a, b) for a, b in zip(again, (x0, mean, xroot, xb_pad)))
    # exact on small integers (d_norm = a power of two, integer mean by construction: one edge per drug)
    if cat:
        xi = torch.randint(-3, 4, (n, ne), generator=g).float()
        bi, ri = torch.randint(-2, 3, basis.shape, generator=g).float(), torch.randint(-2, 3, root.shape, generator=g).float()
        hi, wi = torch.randint(-2, 3, h.shape, generator=g).float(), torch.randint(-2, 3, w.shape, generator=g).float()
        src1 = torch.randint(0, n_src, (n,), generator=g)
        g1 = hier_graph(torch.stack([src1, torch.arange(n) + n_src]).to(DEV), n_src + n, n_src, table_rows=n_src, d=p)
        xb_pad = torch.zeros(n_pad, nb, 32, device=DEV)
        x0i, _, xri = encoder.drug_mix_gather_xb(xi.to(DEV), hi.to(DEV), wi.to(DEV), torch.full((n,), 0.5, device=DEV), cat, g1,
                                                 bi.to(DEV), ri.to(DEV), xb_pad[:, :, :d_out])
        want_x0 = torch.cat([xi.double() * 2, hi.double()[src1] @ wi.double()], 1)
        assert torch.equal(x0i.double().cpu(), want_x0)
        assert torch.equal(xb_pad[:n, :, :d_out].double().cpu(), torch.einsum('nk,bkc->nbc', want_x0, bi.double()))
        assert torch.equal(xri.double().cpu(), want_x0 @ ri.double())


@pytest.mark.parametrize('n,n_src,p,q,ne,cat,c1', [(645, 3640, 16, 16, 48, True, 32), (37, 50, 8, 8, 12, True, 24),
                                                   (130, 64, 16, 64, 64, False, 32), (33, 300, 6, 10, 10, True, 7)])
def test_pd_stage_bwd(ops, n, n_src, p, q, ne, cat, c1):
    """tipk_pd_stage_bwd: d xd, d W_h, the transposed P -> D gather, conv2's g W and the slabs of d W2 / d b2 in ONE launch
    == the fp64 definitions (autograd of src/layers.py:526-539 and of GCNConv 2); source rows without edges, repeated edges,
    a strided upstream gradient, a row scale; bitwise reproducible."""
    from tip_amd import encoder
    from tip_amd.layers import hier_graph
    g = torch.Generator().manual_seed(n + p + q + c1)
    E = 6 * n + 3
    src = torch.randint(0, max(1, n_src - 3), (E,), generator=g)      # the last source rows have no edge
    dst = torch.randint(0, max(1, n - 2), (E,), generator=g)
    src[:4] = src[4:8]; dst[:4] = dst[4:8]
    ei = torch.stack([src, dst + n_src]).to(DEV)
    graph = hier_graph(ei, n_src + n, n_src, table_rows=n_src, d=p)
    cols = ne + q if cat else ne
    gup = torch.randn(n, cols + 3, generator=g)[:, 1:-2]
    mean, w = torch.randn(n, p, generator=g), torch.randn(p, q, generator=g)
    dn = torch.rand(n, generator=g) + 0.5
    agg = torch.randn(n_src, c1, generator=g)
    w2_store = torch.randn(c1, p, generator=g)                         # [in, out] storage behind the [out, in] shape
    rs = torch.rand(n_src, generator=g) + 0.5
    cnt = torch.bincount(dst, minlength=n).clamp(min=1).double()
    gd = gup.double()
    g_pd = gd[:, ne:] if cat else gd
    g_mean = (g_pd @ w.double().t()) / cnt.unsqueeze(1)
    g_h = torch.zeros(n_src, p, dtype=torch.float64).index_add_(0, src, g_mean[dst])
    w2 = w2_store.t()
    want_gw = (g_h @ w2.double()) * rs.double().unsqueeze(1)

    def run(scale):
        return encoder.pd_stage_bwd(gup.to(DEV), dn.to(DEV), mean.to(DEV), w.to(DEV), ne, cat, graph, agg.to(DEV), w2_store.to(DEV).t(),
                                    scale)
    g_xd, j_wh, gw, j_w2, j_b2 = run(rs.to(DEV))
    ops.gemm_group([], [j_w2, j_b2, j_wh])
    g_w = j_wh.out
    close(g_xd, gd[:, :ne] / dn.double().unsqueeze(1), rtol=2e-5, atol=2e-6)
    close(g_w, mean.double().t() @ g_pd, rtol=2e-5, atol=2e-5 * float((mean.double().t() @ g_pd).abs().max()))
    close(gw, want_gw, rtol=2e-5, atol=2e-5 * float(want_gw.abs().max()))
    want_w2 = agg.double().t() @ g_h
    close(j_w2.out, want_w2, rtol=2e-5, atol=2e-5 * float(want_w2.abs().max()))
    close(j_b2.out, g_h.sum(0), rtol=2e-5, atol=2e-5 * float(g_h.abs().sum(0).max()))
    assert float(gw[n_src - 3:].abs().max()) == 0.0                    # rows without edges: exact zeros
    again = run(rs.to(DEV))
    ops.gemm_group([], [again[3], again[4], again[1]])
    assert torch.equal(again[0], g_xd) and torch.equal(again[1].out, g_w) and torch.equal(again[2], gw)
    assert torch.equal(again[3].out, j_w2.out) and torch.equal(again[4].out, j_b2.out)
    plain = run(None)
    close(plain[2], g_h @ w2.double(), rtol=2e-5, atol=2e-5 * float((g_h @ w2.double()).abs().max()))


def test_pair_att_gather_two_tables_one_launch(ops):
    """tipk_stream_gather_parts_two == two tipk_stream_gather_parts launches, bit for bit (BioSNAP-shaped plan)."""
    from tip_amd import encoder
    from tip_amd.plan import build_pair_bwd_plan
    R, N, nb = 300, 200, 32
    g = torch.Generator().manual_seed(5)
    src, dst, rel = _random_dd_graph(N, R, 400, g, True)
    scale = 1.0 / torch.bincount(dst, minlength=N).clamp(min=1).float()
    plan = build_pair_bwd_plan(src, dst, rel, N, R, scale, True, 32, nb // 4, ops.rel_stream_piece()).to(DEV)
    pa = torch.randn(2 * plan.n_alloc + 1, nb, generator=g).to(DEV)
    pb_ = torch.randn(2 * plan.n_alloc + 1, nb, generator=g).to(DEV)
    ja, jb = encoder.pair_att_gather_two(plan, pa, pb_)
    ra, rb = ops.pair_att_gather(plan, pa), ops.pair_att_gather(plan, pb_)
    ops.gemm_group([], [ja, jb, ra, rb])
    assert torch.equal(ja.out, ra.out) and torch.equal(jb.out, rb.out)
