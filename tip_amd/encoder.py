"""The encoder's training step as ONE autograd node with an explicit launch schedule (round 6).

`FMEncoder.forward` (src/layers.py:520-550) used to be five `torch.autograd.Function`s (GCNConv x 2, the P -> D stage, the two
R-GCN layers) that handed work to each other through attributes on tensors and graphs: a slab sum left "pending" for the next
layer, a token telling layer 2 that layer 1 had gathered its pair cells, a link object carrying bias partials backwards.  The
per-layer nodes also fixed WHERE a launch could sit: work that only needs parameters, or that two layers could share, had to
stay inside its layer's node.  Here the whole pass is one node, `_EncoderStep`, whose forward and backward are plain launch
lists over libtipk (include/tipk.h) -- 8 + 8 launches at BioSNAP, where the per-layer nodes made 9 + 11:

    forward   1  conv1 = gather_sum (A_hat W1^T + b1, ReLU)                                      tipk_gather_sum
              2  conv2 on the kept rows: gather + dense map                                       tipk_gather_sum_lin
              3  P -> D mean, map, /d_norm, cat | add AND XB1 = x0 basis1, x0 root1               tipk_drug_mix_gather_xb_fwd
              4  pair cells of BOTH layers                                                        tipk_stream_gather_two
              5  pair product, layer 1                                                            tipk_pair_product
              6  slab sum + 1/deg + x0 root1 + ReLU -> x1, XB2 = x1 basis2, x1 root2              tipk_sum_slabs_xb
              7  pair product, layer 2                                                            tipk_pair_product
              8  slab sum + 1/deg + x1 root2 -> z                                                 tipk_sum_slabs_ex
    backward  9  dXB2 + pair-gradient rows, layer 2                                               tipk_rgcn_pair_grads
             10  d basis2, d root2, dX1 (ReLU gate)                                               tipk_gemm_wg_group
             11  dXB1 + pair-gradient rows, layer 1                                               tipk_rgcn_pair_grads
             12  d att slabs of BOTH layers                                                       tipk_stream_gather_parts_two
             13  d basis1, d root1, dX0 + the two d att slab sums                                 tipk_gemm_wg_group
             14  d embed, d W_h, transposed P -> D gather, conv2's g W and d W2 / d b2 partials   tipk_pd_stage_bwd
             15  conv2's transposed gather + conv1's ReLU gate + d b1 partials + d W2 / d b2 sums tipk_gather_sum_riders
             16  conv1's transposed gather = d W1 (+ the d b1 sum)                                tipk_gather_sum_riders

Nothing is handed over through tensor attributes; what the backward pass needs is saved on the node (N x d activations, the
graphs' plans), and the pair buffers (cells, node-major XB) stay with their graph, stamped as before: a backward pass that finds
another stamp than its forward pass left recomputes them.

`usable(...)` says whether a call can take this node (identity protein features, a pair-form D-D graph for both layers, widths
the fused kernels take, no relation sharding); everything else runs on the per-layer nodes of tip_amd/ops.py, unchanged.
"""
import torch

from . import ops
from ._lib import check, lib, ptr, stream_ptr


class EncoderPlans(object):
    """What `_EncoderStep` needs besides tensors: the graphs of the five stages (built and cached by the layers)."""
    __slots__ = ('pp', 'pp_rows', 'pd', 'dd1', 'dd2', 'cat')

    def __init__(self, pp, pp_rows, pd, dd1, dd2, cat):
        self.pp, self.pp_rows, self.pd, self.dd1, self.dd2, self.cat = pp, pp_rows, pd, dd1, dd2, cat


def _pair_ok(graph, att, basis, n):
    """The pair form applies to this layer exactly as `_RGCN.forward` decides it (tip_amd/ops.py)."""
    r, nb = att.shape
    d_out = basis.shape[2]
    pair = graph.pair_fwd if r > 0 else None
    if pair is None:
        return False
    if pair.symmetric and not lib().tipk_pair_product_supported(nb, d_out):
        return False
    split = ops.stream_gather_split(r, nb)
    if not (pair.n_table == r and pair.n_rows == n * n and split and (nb // split) // 4 == pair.lanes):
        return False
    return ops.pair_grads_supported(nb, d_out) and graph.pair_bwd is not None


def usable(plans, xd, w_h, d_norm, c2_weight, c1_bias, c2_bias, basis1, att1, basis2, att2):
    """True if the fused schedule takes this call; decided on shapes, plans and kernel support queries only."""
    if plans.pp is None or plans.pp_rows is None or plans.pd is None or plans.dd1 is None or plans.dd2 is None:
        return False
    if c1_bias is None or c2_bias is None or not xd.is_cuda:
        return False
    n, ne = xd.shape
    p, q = w_h.shape
    nb, d_in, d1 = basis1.shape
    _, d1b, d2 = basis2.shape
    cols = ne + q if plans.cat else ne
    L = lib()
    if d_in != cols or d1b != d1 or basis2.shape[0] != nb or att1.shape != att2.shape:
        return False
    if not (w_h.is_contiguous() and (d_norm is None or d_norm.is_contiguous()) and basis1.is_contiguous()):
        return False
    if not L.tipk_drug_mix_gather_xb_supported(int(p), int(q), int(ne), int(plans.cat), int(nb), int(d1)):
        return False
    c1 = c2_weight.shape[1]
    if not (c2_weight.t().is_contiguous() and not c2_weight.is_contiguous()):      # d W2 leaves as [in, out] slabs: the stored layout
        return False
    if not L.tipk_pd_stage_bwd_supported(int(p), int(q), int(n), int(c1)) or 't_wg' not in plans.pd.pd_csr:
        return False
    if not (ops.gather_sum_lin_supported(c1, p, plans.pp_rows.fwd.group_slots) and ops.gather_sum_epilogue_supported(plans.pp_rows.bwd, c1)
            and ops.gather_sum_epilogue_supported(plans.pp.bwd, c1)):
        return False
    if not (_pair_ok(plans.dd1, att1, basis1, n) and _pair_ok(plans.dd2, att2, basis2, n)):
        return False
    if not ops.pair_cells_partner_ok(plans.dd1, att1, plans.dd2, att2, n):
        return False
    if plans.dd1.pair_bwd is not plans.dd2.pair_bwd:                              # one plan, two gradient tables
        return False
    return ops.sum_slabs_xb_supported(d1, d2)


def drug_mix_gather_xb(xd, h, w_h, d_norm, cat, pd_graph, basis, root, xb_nb):
    """(x0, mean, x0 root): the P -> D stage + drug mix + the first R-GCN layer's row-local products, one launch
    (`tipk_drug_mix_gather_xb_fwd`); XB goes to `xb_nb` (the graph's node-major buffer, [N_pad, bases, :d_out] of 32-column rows)."""
    csr = pd_graph.pd_csr
    n, ne = xd.shape
    p, q = w_h.shape
    nb, cols, d_out = basis.shape
    assert h.shape == (csr['n_src'], p) and csr['fwd_ptr'].numel() == n + 1 and h.stride(1) == 1 and xd.stride(1) == 1
    assert xb_nb.stride()[-2:] == (32, 1) and xb_nb.shape[1] == nb and xb_nb.stride(0) == nb * 32 and xb_nb.shape[0] >= n
    out = torch.empty((n, cols), dtype=torch.float32, device=xd.device)
    mean = torch.empty((n, p), dtype=torch.float32, device=xd.device)
    xroot = torch.empty((n, d_out), dtype=torch.float32, device=xd.device)
    with ops._timed('drug_mix_gather_xb_fwd[%dx%dx%d -> %dx%d]' % (n, p, q, nb, d_out)):
        check(lib().tipk_drug_mix_gather_xb_fwd(ptr(xd), xd.stride(0), ptr(d_norm), ptr(h), h.stride(0), ptr(csr['fwd_ptr']),
                                                ptr(csr['fwd_src']), ptr(csr['scale']), ptr(csr['fwd_wg']), ptr(csr['fwd_order']),
                                                csr['fwd_wg'].shape[0], ptr(w_h), p, q, n, ne, int(cat), ptr(out), out.stride(0), ptr(mean),
                                                ptr(basis), ptr(root), nb, d_out, ptr(xb_nb), ptr(xroot), stream_ptr(xd.device)),
              'tipk_drug_mix_gather_xb_fwd')
    return out, mean, xroot


def pd_stage_bwd(g_x0, d_norm, mean, w_h, ne, cat, pd_graph, agg, w2, row_scale, want_xd=True):
    """(d xd, slab job of d W_h, gw = (g_h W2) * row_scale, slab job of d W2 ([in, out] storage), slab job of d b2): the backward pass of the
    P -> D stage down to the input of conv2's transposed gather, one launch (`tipk_pd_stage_bwd`)."""
    csr = pd_graph.pd_csr
    n, p = mean.shape
    q = w_h.shape[1]
    n_src, c1 = agg.shape
    assert w2.shape == (p, c1) and g_x0.stride(1) == 1 and agg.stride(1) == 1 and csr['t_ptr'].numel() == n_src + 1
    dev = g_x0.device
    n_slabs = int(csr['t_wg'].numel()) - 1
    g_xd = torch.empty((n, ne), dtype=torch.float32, device=dev) if want_xd else None
    g_w = torch.empty((int(lib().tipk_pd_stage_bwd_wh_slabs()), p, q), dtype=torch.float32, device=dev)
    gw = torch.empty((n_src, c1), dtype=torch.float32, device=dev)
    dw2 = torch.empty((n_slabs, c1, p), dtype=torch.float32, device=dev)
    db2 = torch.empty((n_slabs, p), dtype=torch.float32, device=dev)
    with ops._timed('pd_stage_bwd[%dx%dx%d,rows=%d]' % (n, p, q, n_src)):
        check(lib().tipk_pd_stage_bwd(ptr(g_x0), g_x0.stride(0), ptr(d_norm), ptr(mean), ptr(w_h), p, q, n, ne, int(cat),
                                      ptr(g_xd), g_xd.stride(0) if g_xd is not None else 0, ptr(g_w),
                                      ptr(csr['t_ptr']), ptr(csr['t_dst']), ptr(csr['t_w']), n_src, ptr(csr['t_wg']), n_slabs,
                                      ptr(agg), agg.stride(0), c1, ptr(w2), w2.stride(0), w2.stride(1), ptr(row_scale),
                                      ptr(gw), gw.stride(0), ptr(dw2), ptr(db2), stream_ptr(dev)), 'tipk_pd_stage_bwd')
    return g_xd, ops.slab_job(g_w), gw, ops.slab_job(dw2), ops.slab_job(db2)


def pair_att_gather_two(pb, pg_a, pg_b):
    """The d att slab jobs of two layers on one pair-backward plan, one launch (`tipk_stream_gather_parts_two`)."""
    nb = pg_a.shape[1]
    assert pg_a.shape == pg_b.shape == (2 * pb.n_alloc + 1, nb) and pg_a.is_contiguous() and pg_b.is_contiguous()
    gp = pb.gather
    sa = torch.empty((pb.n_parts, pb.n_rel, nb), dtype=torch.float32, device=pg_a.device)
    sb = torch.empty_like(sa)
    with ops._timed('pair_att_gather2[parts=%d,edges=%d]' % (pb.n_parts, gp.n_edges)):
        check(lib().tipk_stream_gather_parts_two(ptr(pg_a), ptr(pg_b), nb, nb, pb.n_alloc, ptr(pb.part_first), pb.part_len,
                                                 ptr(pb.wg_part), gp.n_wg, ptr(gp.wave_ptr), ptr(gp.cells), ptr(gp.ids), gp.idx_unit,
                                                 ptr(gp.zero_ptr), ptr(gp.zero_rows), ptr(sa), ptr(sb), nb, stream_ptr(pg_a.device)),
              'tipk_stream_gather_parts_two')
    return ops.slab_job(sa), ops.slab_job(sb)


def rgcn_dense_backward(x, basis, root, g, g_xb, gate_x, slab_jobs):
    """(dX, d basis, d root) of an R-GCN layer given dXB [bases, N, d_out] and g [N, d_out] (1/deg not applied to the root
    term), plus the pending ordered slab sums `slab_jobs` (d att) finished in the same launch where the reductions fit one
    workgroup per tile (`tipk_gemm_wg_group`); otherwise the grouped split-K products of `_RGCN.backward`."""
    n, d_in = x.shape
    nb = basis.shape[0]
    slab_jobs = list(slab_jobs)
    if len(slab_jobs) <= ops._lib.WG_SUMS_MAX:
        w_basis = ops.wg_gemm_job(x.t(), g_xb)
        w_root = ops.wg_gemm_job(x.t(), g)
        w_x = ops.wg_gemm_job(g_xb, basis.transpose(1, 2), reduce_batch=True, a2=g, b2=root.t(), gate=gate_x)
        if w_basis is not None and w_root is not None and w_x is not None:
            ops.wg_gemm_group([w_basis, w_root, w_x], slab_jobs)
            return w_x.out, w_basis.out, w_root.out
    j_root = ops.gemm_job(x.t(), g)
    j_basis = ops.gemm_job(x.t(), g_xb)
    j_xr = ops.gemm_job(g, root.t(), ksplit=1)
    g_x = j_xr.out
    j_xq = ops.gemm_job(g_xb, basis.transpose(1, 2), out=g_x, c_in=g_x, reduce_batch=True, kgroup=ops.large_kgroup(n, nb))
    if j_xq.slabs is not None:
        if gate_x is not None:
            j_xq.gate = gate_x
        ops.gemm_group([j_basis, j_root, j_xr, j_xq], slab_jobs)
    else:
        ops.gemm_group([j_basis, j_root, j_xr], slab_jobs)
        ops.gemm_group([j_xq])
        if gate_x is not None:
            g_x = ops.rows_affine(g_x, gate=gate_x)
    return g_x, j_basis.out, j_root.out


class _EncoderStep(torch.autograd.Function):
    """z = FMEncoder.forward(...) for identity protein features on a pair-form D-D graph (module docstring)."""

    @staticmethod
    def forward(ctx, xd, w1, b1, w2, b2, w_h, d_norm, basis1, att1, root1, basis2, att2, root2, plans):
        f32c = ops._f32c
        xd, w_h = f32c(xd), f32c(w_h).contiguous()
        basis1, att1, root1 = basis1.contiguous(), att1.contiguous(), root1.contiguous()
        basis2, att2, root2 = basis2.contiguous(), att2.contiguous(), root2.contiguous()
        n = xd.shape[0]
        nb, _, d1 = basis1.shape
        d2 = basis2.shape[2]
        dev = xd.device
        g1, g2 = plans.dd1, plans.dd2
        # 1. conv1 on identity features: lin(I) = W1^T is the parameter's own storage (tip_amd.layers._Lin)
        xl = ops.transpose(w1)
        h1 = ops.gather_sum(plans.pp.fwd, xl, row_scale=plans.pp.scale, bias=b1, relu=True)
        # 2. conv2, aggregate first, on the rows the P -> D stage reads
        agg2, h_prot = ops.gather_sum_lin(plans.pp_rows.fwd, h1, w2, b2, False, row_scale=plans.pp_rows.scale)
        # 3. P -> D + mix + layer 1's row-local products
        cells1, xb1, zeros1 = g1.pair_buffers(n, nb, d1, dev)
        cells2, xb2, zeros2 = g2.pair_buffers(n, nb, d2, dev)
        x0, mean, xroot1 = drug_mix_gather_xb(xd, h_prot, w_h, d_norm, plans.cat, plans.pd, basis1, root1, xb1)
        # 4. the pair cells of both layers (they depend on att alone)
        pair = g1.pair_fwd
        ops.stream_gather_two(pair, att1, att2, cells1.view(-1, nb)[:n * n], cells2.view(-1, nb)[:n * n])
        g1.pair_stamp += 1
        g2.pair_stamp += 1
        # 5. - 8. products and ordered slab sums
        slabs1 = ops.pair_product(cells1, xb1, symmetric=pair.symmetric, links=getattr(pair, 'links', None), zeros=zeros1)
        x1 = torch.empty((n, d1), dtype=torch.float32, device=dev)
        xroot2 = ops.sum_slabs_xb(slabs1.view(-1, n, d1), g1.scale, xroot1, True, x1, basis2, root2, xb2)
        pair2 = g2.pair_fwd
        slabs2 = ops.pair_product(cells2, xb2, symmetric=pair2.symmetric, links=getattr(pair2, 'links', None), zeros=zeros2)
        z = ops.sum_slabs(slabs2.view(-1, n, d2), row_scale=g2.scale, addend=xroot2, relu=False)
        ctx.plans, ctx.stamps, ctx.ne = plans, (g1.pair_stamp, g2.pair_stamp), xd.shape[1]
        ctx.save_for_backward(w1, w2, w_h, d_norm, basis1, att1, root1, basis2, att2, root2, h1, agg2, mean, x0, x1)
        return z

    @staticmethod
    def backward(ctx, g):
        w1, w2, w_h, d_norm, basis1, att1, root1, basis2, att2, root2, h1, agg2, mean, x0, x1 = ctx.saved_tensors
        plans = ctx.plans
        g1, g2 = plans.dd1, plans.dd2
        g = ops._f32c(g).contiguous()
        n = x0.shape[0]
        nb, _, d1 = basis1.shape
        d2 = basis2.shape[2]
        dev = g.device
        cells1, xb1, _ = g1.pair_buffers(n, nb, d1, dev)
        cells2, xb2, _ = g2.pair_buffers(n, nb, d2, dev)
        if (g1.pair_stamp, g2.pair_stamp) != ctx.stamps:
            # another forward pass has rewritten the graphs' buffers: the same values again (and new stamps, so that THAT
            # pass's backward pass recomputes its own as well)
            ops.gemm(x0, basis1, out=xb1[:n].permute(1, 0, 2))
            ops.gemm(x1, basis2, out=xb2[:n].permute(1, 0, 2))
            ops.stream_gather_two(g1.pair_fwd, att1, att2, cells1.view(-1, nb)[:n * n], cells2.view(-1, nb)[:n * n])
            g1.pair_stamp += 1
            g2.pair_stamp += 1
            ctx.stamps = (g1.pair_stamp, g2.pair_stamp)
        pb = g1.pair_bwd
        # 9. / 10. layer 2: dXB2 and the gradient rows of the linked pairs; then the dense gradients (dX1 gated by x1 > 0)
        pg2, dxb2 = ops.pair_grads(pb, cells2, xb2, g, table=1)
        g_x1, g_basis2, g_root2 = rgcn_dense_backward(x1, basis2, root2, g, dxb2, x1, [])
        # 11. layer 1
        pg1, dxb1 = ops.pair_grads(pb, cells1, xb1, g_x1, table=0)
        # 12. d att of both layers: one launch over the shared plan
        j_att1, j_att2 = pair_att_gather_two(pb, pg1, pg2)
        # 13. layer 1's dense gradients + both d att slab sums
        g_x0, g_basis1, g_root1 = rgcn_dense_backward(x0, basis1, root1, g_x1, dxb1, None, [j_att1, j_att2])
        # 14. the P -> D stage down to conv2's g W, d W2 / d b2 as slabs
        g_xd, j_wh, gw, j_w2, j_b2 = pd_stage_bwd(g_x0, d_norm, mean, w_h, ctx.ne, plans.cat, plans.pd, agg2, w2, plans.pp_rows.scale,
                                                 want_xd=ctx.needs_input_grad[0])
        # 15. conv2's transposed gather; conv1's ReLU gate and the partial rows of its bias gradient in the epilogue
        g_h1, parts = ops.gather_sum(plans.pp_rows.bwd, gw, riders=[j_w2, j_b2, j_wh], gate=h1, colsum=True)
        # 16. conv1's transposed gather IS d W1 (identity features)
        g_agg = ops.rows_affine(g_h1, row_mul=plans.pp.scale) if plans.pp.scale is not None else g_h1
        j_b1 = ops.slab_job(parts)
        g_table = ops.gather_sum(plans.pp.bwd, g_agg, riders=[j_b1])
        g_w1 = g_table.t() if w1.t().is_contiguous() else ops.transpose(g_table)
        g_w2 = j_w2.out.t()                                                   # [in, out] storage behind the [out, in] shape
        return (g_xd, g_w1, j_b1.out.view(-1), g_w2, j_b2.out.view(-1), j_wh.out, None,
                g_basis1, j_att1.out, g_root1, g_basis2, j_att2.out, g_root2, None)


class _PDStage(torch.autograd.Function):
    """x0 = mix(xd / d_norm, mean_targets(conv2(h1)) W_h) as ONE autograd node (conv2 aggregate-first on the kept rows + the P -> D
    stage): the part of the encoder step that a relation-SHARDED run repeats on every rank -- there the R-GCN layers stay on
    their own nodes (`ops._RGCN` with the shard's collectives), but this stage can use the fused launches of `_EncoderStep`:
    forward `tipk_gather_sum_lin` + `tipk_drug_mix_gather_fwd`, backward `tipk_pd_stage_bwd` + conv2's transposed gather with
    conv1's ReLU gate and bias partials in its epilogue (3 + 2 launches where the per-layer nodes take 3 + 4).  The returned
    gradient of h1 is already masked with (h1 > 0); the partial rows of conv1's bias gradient go to `link.parts`."""

    @staticmethod
    def forward(ctx, h1, w2, b2, xd, w_h, d_norm, pp_rows, pd, cat, link):
        h1, xd, w_h = ops._f32c(h1), ops._f32c(xd), ops._f32c(w_h).contiguous()
        agg2, h_prot = ops.gather_sum_lin(pp_rows.fwd, h1, w2, b2, False, row_scale=pp_rows.scale)
        x0, mean, w_h = ops.drug_mix_gather_launch(xd, h_prot, w_h, d_norm, cat, pd)
        ctx.pp_rows, ctx.pd, ctx.cat, ctx.link, ctx.ne = pp_rows, pd, cat, link, xd.shape[1]
        ctx.save_for_backward(h1, w2, w_h, d_norm, agg2, mean)
        return x0

    @staticmethod
    def backward(ctx, g):
        h1, w2, w_h, d_norm, agg2, mean = ctx.saved_tensors
        pd = ctx.pd
        g = ops._f32c(g)
        g_xd, j_wh, gw, j_w2, j_b2 = pd_stage_bwd(g, d_norm, mean, w_h, ctx.ne, ctx.cat, pd, agg2, w2, ctx.pp_rows.scale,
                                                 want_xd=ctx.needs_input_grad[3])
        want = ctx.link is not None
        res = ops.gather_sum(ctx.pp_rows.bwd, gw, riders=[j_w2, j_b2, j_wh], gate=h1, colsum=want)
        g_h1 = res[0] if want else res
        if want:
            ctx.link.parts = res[1]
        return g_h1, j_w2.out.t(), j_b2.out.view(-1), g_xd, j_wh.out, None, None, None, None, None


def pd_stage_usable(pp_rows, pd, xd, w_h, d_norm, w2, b2):
    """True if `pd_stage` takes this call (the same support queries as `usable`, without the D-D layers)."""
    if pp_rows is None or pd is None or b2 is None or not xd.is_cuda or 't_wg' not in pd.pd_csr:
        return False
    p, q = w_h.shape
    c1 = w2.shape[1]
    if not (w2.t().is_contiguous() and not w2.is_contiguous() and w_h.is_contiguous() and d_norm is not None and d_norm.is_contiguous()):
        return False
    if not lib().tipk_pd_stage_bwd_supported(int(p), int(q), int(xd.shape[0]), int(c1)):
        return False
    return bool(ops.drug_mix_gather_supported(torch.empty((1, p), device=xd.device), w_h, d_norm)
                and ops.gather_sum_lin_supported(c1, p, pp_rows.fwd.group_slots) and ops.gather_sum_epilogue_supported(pp_rows.bwd, c1))


def pd_stage(h1, w2, b2, xd, w_h, d_norm, pp_rows, pd, cat, link=None):
    return _PDStage.apply(h1, w2, b2, xd, w_h, d_norm, pp_rows, pd, cat, link)


def encoder_step(xd, w1, b1, w2, b2, w_h, d_norm, basis1, att1, root1, basis2, att2, root2, plans):
    return _EncoderStep.apply(xd, w1, b1, w2, b2, w_h, d_norm, basis1, att1, root1, basis2, att2, root2, plans)
