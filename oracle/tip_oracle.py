"""ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (plain torch-CPU / numpy, explicit forward AND backward, no autograd, no HIP) of
the TIP hot path named by BASELINE.json `north_star`.  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import this file; nothing under `tip_amd/` does.

Every function cites the reference lines it restates (paths relative to /root/reference).  The
gather / scatter-mean / GCN-normalisation arithmetic lives in PyG 2.0.1 + torch-scatter 2.0.8
(`environment_tip_gpu.yml:69,79`), which are not vendored in the reference and not installed here;
their published semantics are restated (see `oracle/pyg_restated`).

Pinning: the reference has no tests or golden vectors for this path, so the
oracle is pinned by `tests/golden/*.npz` -- outputs and parameter gradients of the reference's
*own* `src/layers.py`, imported unchanged in the build container over `oracle/pyg_restated` by
`oracle/make_golden.py` (committed).  `tests/test_oracle_golden.py` checks this file against them.
The PyG boundary itself stays "parity unpinned" (DESIGN.md).

All functions take/return CPU tensors; dtype follows the inputs (fp32 like the reference, or fp64
when a test wants a tight reference).  Index tensors are int64 as in the reference.
"""
import math

import numpy as np
import torch

EPS = 1e-13          # src/layers.py:15


# ---------------------------------------------------------------------------------------------
# helpers: the two PyG/torch-scatter primitives everything else is made of
# ---------------------------------------------------------------------------------------------
def scatter_add_rows(src_rows, index, n_rows):
    """torch_scatter.scatter(src, index, dim=0, dim_size=n, reduce='sum')."""
    out = torch.zeros((n_rows, src_rows.shape[1]), dtype=src_rows.dtype)
    out.index_add_(0, index, src_rows)
    return out


def in_degree(index, n_rows, dtype):
    """count used by reduce='mean': number of incoming messages, clamped to >= 1."""
    cnt = torch.bincount(index, minlength=n_rows).to(dtype)
    return cnt.clamp_(min=1)


def gather_sum(table, row_id, dst, n_rows, weight=None, chunk=1 << 20):
    """out[d] = sum_{e: dst[e]=d} weight[e] * table[row_id[e]]  (chunked: no E x d temporary)."""
    out = torch.zeros((n_rows, table.shape[1]), dtype=table.dtype)
    for a in range(0, row_id.numel(), chunk):
        rows = table.index_select(0, row_id[a:a + chunk])
        if weight is not None:
            rows = rows * weight[a:a + chunk].unsqueeze(1)
        out.index_add_(0, dst[a:a + chunk], rows)
    return out


# ---------------------------------------------------------------------------------------------
# A1  P-P encoder: two GCNConv layers  (src/layers.py:380-395; GCNConv = PyG 2.0.1)
# ---------------------------------------------------------------------------------------------
def gcn_norm(edge_index, num_nodes, dtype=torch.float32):
    """PyG 2.0.1 `gcn_norm(add_self_loops=True, improved=False)`: drop existing self loops, add
    one loop (weight 1) per node, deg = in-degree incl. loop, norm = d^-1/2[row] d^-1/2[col]."""
    row, col = edge_index[0], edge_index[1]
    keep = row != col
    loop = torch.arange(num_nodes, dtype=row.dtype)
    row = torch.cat([row[keep], loop])
    col = torch.cat([col[keep], loop])
    deg = torch.bincount(col, minlength=num_nodes).to(dtype)
    dis = deg.pow(-0.5)
    dis[torch.isinf(dis)] = 0
    return row, col, dis[row] * dis[col]


def gcn_conv_fwd(x_lin, row, col, norm, bias):
    """GCNConv after the linear map: out = scatter_add(norm * x_lin[row] -> col) + bias."""
    return gather_sum(x_lin, row, col, x_lin.shape[0], norm) + bias


def gcn_conv_bwd(g_out, row, col, norm):
    """Gradients of gcn_conv_fwd w.r.t. (x_lin, bias): transpose aggregation, column sum."""
    return gather_sum(g_out, col, row, g_out.shape[0], norm), g_out.sum(0)


def pp_encoder_fwd(w1, b1, w2, b2, pp_edge_index, n_prot, x=None):
    """PPEncoder.forward (src/layers.py:391-395).  `x=None` means identity features (the
    reference feeds `sparse_id(n_prot)`, prepare.py:23), so `lin1(x) = w1.T`.
    w1 [hid1, n_prot], w2 [hid2, hid1] are the `lin.weight`s.  Returns (h2, saved)."""
    row, col, norm = gcn_norm(pp_edge_index, n_prot, w1.dtype)
    xl1 = w1.t() if x is None else x @ w1.t()
    a1 = gcn_conv_fwd(xl1, row, col, norm, b1)
    h1 = torch.relu(a1)
    xl2 = h1 @ w2.t()
    h2 = gcn_conv_fwd(xl2, row, col, norm, b2)
    return h2, (row, col, norm, h1, x)


def pp_encoder_bwd(g_h2, w1, w2, saved):
    """Returns (g_w1, g_b1, g_w2, g_b2)."""
    row, col, norm, h1, x = saved
    g_xl2, g_b2 = gcn_conv_bwd(g_h2, row, col, norm)
    g_w2 = g_xl2.t() @ h1
    g_h1 = g_xl2 @ w2
    g_a1 = g_h1 * (h1 > 0).to(g_h1.dtype)
    g_xl1, g_b1 = gcn_conv_bwd(g_a1, row, col, norm)
    g_w1 = g_xl1.t() if x is None else g_xl1.t() @ x
    return g_w1, g_b1, g_w2, g_b2


# ---------------------------------------------------------------------------------------------
# A2  MyHierarchyConv  (src/layers.py:196-247)
# ---------------------------------------------------------------------------------------------
def hier_conv_fwd(x_all, edge_index, weight, n_source):
    """mean-aggregate x_all[src] at dst over the concatenated node space, keep rows
    [n_source:], multiply by weight (:229-242; bias is None in every reference use)."""
    n_all = x_all.shape[0]
    src, dst = edge_index[0], edge_index[1]
    cnt = in_degree(dst, n_all, x_all.dtype)
    mean = gather_sum(x_all, src, dst, n_all) / cnt.unsqueeze(1)
    tail = mean[n_source:]
    return tail @ weight, (tail, cnt)


def hier_conv_bwd(g_out, x_all_rows, edge_index, weight, n_source, saved):
    """Returns (g_x_all, g_weight)."""
    tail, cnt = saved
    src, dst = edge_index[0], edge_index[1]
    g_w = tail.t() @ g_out
    g_mean = torch.zeros((x_all_rows, weight.shape[0]), dtype=g_out.dtype)
    g_mean[n_source:] = g_out @ weight.t()
    g_mean = g_mean / cnt.unsqueeze(1)
    return gather_sum(g_mean, dst, src, x_all_rows), g_w


# ---------------------------------------------------------------------------------------------
# A4/A5  R-GCN layer with basis decomposition and global-mean aggregation
#        (MyRGCNConv2 src/layers.py:102-193, MyRGCNConv :21-99 -- same arithmetic)
# ---------------------------------------------------------------------------------------------
def rgcn_fwd(x, edge_index, range_list, basis, att, root):
    """out = mean_dst( concat_r X[src_r] W_r ) + X root,  W = att @ basis  (:162-188).

    Evaluated basis-first / transform-then-gather: XB_b = X basis_b,
    Y_r = sum_b att[r,b] XB_b, agg[d] = sum_{(s->d) in r} Y_r[s]; the mean denominator is the
    in-degree over ALL relations clamped to >= 1 (torch-scatter 'mean').
    Returns (out, saved)."""
    n, r = x.shape[0], att.shape[0]
    d_out = basis.shape[2]
    src, dst = edge_index[0], edge_index[1]
    rel = _relation_of_edges(range_list, src.numel())
    xb = torch.einsum('ni,bio->bno', x, basis)                    # [B, N, out]
    y = (att @ xb.reshape(att.shape[1], -1)).reshape(r * n, d_out)
    deg = in_degree(dst, n, x.dtype)
    agg = gather_sum(y, rel * n + src, dst, n)
    out = agg / deg.unsqueeze(1) + x @ root
    return out, (xb, deg, rel)


def rgcn_bwd(g_out, x, edge_index, basis, att, root, saved):
    """Returns (g_x, g_basis, g_att, g_root)."""
    xb, deg, rel = saved
    n, r = x.shape[0], att.shape[0]
    nb, d_in, d_out = basis.shape
    src, dst = edge_index[0], edge_index[1]
    g_root = x.t() @ g_out
    g_x = g_out @ root.t()
    g_agg = g_out / deg.unsqueeze(1)
    g_y = gather_sum(g_agg, dst, rel * n + src, r * n).reshape(r, n * d_out)   # dY_r = A_r^T G'
    g_att = g_y @ xb.reshape(nb, -1).t()
    g_xb = (att.t() @ g_y).reshape(nb, n, d_out)
    g_basis = torch.einsum('ni,bno->bio', x, g_xb)
    g_x = g_x + torch.einsum('bno,bio->ni', g_xb, basis)
    return g_x, g_basis, g_att, g_root


def _relation_of_edges(range_list, n_edges):
    """relation id per edge from the [R,2] (start,end) table (`dd_train_range`)."""
    rg = range_list.to(torch.int64)
    sizes = rg[:, 1] - rg[:, 0]
    assert int(sizes.sum()) == n_edges and bool((rg[1:, 0] == rg[:-1, 1]).all()) and int(rg[0, 0]) == 0
    return torch.repeat_interleave(torch.arange(rg.shape[0]), sizes)


def rgcn_fwd_reference_shaped(x, edge_index, range_list, basis, att, root):
    """The op sequence PyG 2.0.1 + src/layers.py:157-188 execute, under autograd: lift
    x_j = x[src] (E x in), per-relation slice + mm, cat (E x out), scatter mean, + x root.
    This is the 'PyG-CPU path' flavour of the CPU baseline."""
    nb, d_in, d_out = basis.shape
    w = (att @ basis.view(nb, -1)).view(att.shape[0], d_in, d_out)
    x_j = x.index_select(0, edge_index[0])
    parts = []
    for et in range(range_list.shape[0]):
        a, b = int(range_list[et, 0]), int(range_list[et, 1])
        parts.append(x_j[a:b, :] @ w[et])
    msg = torch.cat(parts)
    n = x.shape[0]
    agg = torch.zeros((n, d_out), dtype=x.dtype).index_add_(0, edge_index[1], msg)
    cnt = torch.bincount(edge_index[1], minlength=n).to(x.dtype).clamp_(min=1)
    return agg / cnt.unsqueeze(1) + x @ root


# ---------------------------------------------------------------------------------------------
# A3 + composition: FMEncoder.forward  (src/layers.py:520-550)
# ---------------------------------------------------------------------------------------------
def _nonidentity_features(data, embed):
    f = data.get('d_feat') if isinstance(data, dict) else None
    if f is None or not f.is_sparse or f.shape[0] == f.shape[1]:
        return None
    return f.to(embed.dtype)


def fm_encoder_fwd(p, data, mod='cat'):
    """p: dict of parameters under the reference's state_dict names (without 'encoder.'):
    embed, pp_encoder.conv{1,2}.lin.weight/.bias, hgcn.weight, rgcn{1,2}.{basis,att,root}.
    data: dict with dd_train_idx, dd_train_range, d_norm, pp_train_indices, dp_edge_index,
    n_drug, n_prot; drug features are the identity unless data['d_feat'] is a NON-identity sparse
    matrix (the mono side-effect features of data/utils.py:117-132), protein features the identity.
    Returns (z, saved)."""
    n_prot, n_drug = data['n_prot'], data['n_drug']
    h2, s_pp = pp_encoder_fwd(p['pp_encoder.conv1.lin.weight'], p['pp_encoder.conv1.bias'],
                              p['pp_encoder.conv2.lin.weight'], p['pp_encoder.conv2.bias'],
                              data['pp_train_indices'], n_prot)
    x_all = torch.cat([h2, torch.zeros((n_drug, h2.shape[1]), dtype=h2.dtype)])          # :526
    pd, s_h = hier_conv_fwd(x_all, data['dp_edge_index'], p['hgcn.weight'], n_prot)       # :528
    emb = p['embed']
    feat = _nonidentity_features(data, emb)
    if feat is not None:
        emb = torch.sparse.mm(feat, emb)                                                  # :532 torch.matmul(x_drug, embed)
    xd = emb / data['d_norm'].to(emb.dtype).view(-1, 1)                                   # :534
    x0 = torch.cat([xd, pd], dim=1) if mod == 'cat' else xd + pd                          # :536-539
    a1, s_r1 = rgcn_fwd(x0, data['dd_train_idx'], data['dd_train_range'],
                        p['rgcn1.basis'], p['rgcn1.att'], p['rgcn1.root'])                # :545
    x1 = torch.relu(a1)                                                                   # :547
    z, s_r2 = rgcn_fwd(x1, data['dd_train_idx'], data['dd_train_range'],
                       p['rgcn2.basis'], p['rgcn2.att'], p['rgcn2.root'])                 # :548
    return z, dict(pp=s_pp, h=s_h, r1=s_r1, r2=s_r2, x0=x0, x1=x1, x_all_rows=x_all.shape[0])


def fm_encoder_bwd(g_z, p, data, saved, mod='cat'):
    """Gradients for every parameter, keyed like `p`."""
    g = {}
    ei = data['dd_train_idx']
    g_x1, g['rgcn2.basis'], g['rgcn2.att'], g['rgcn2.root'] = rgcn_bwd(
        g_z, saved['x1'], ei, p['rgcn2.basis'], p['rgcn2.att'], p['rgcn2.root'], saved['r2'])
    g_a1 = g_x1 * (saved['x1'] > 0).to(g_x1.dtype)
    g_x0, g['rgcn1.basis'], g['rgcn1.att'], g['rgcn1.root'] = rgcn_bwd(
        g_a1, saved['x0'], ei, p['rgcn1.basis'], p['rgcn1.att'], p['rgcn1.root'], saved['r1'])
    n_e = p['embed'].shape[1]
    if mod == 'cat':
        g_xd, g_pd = g_x0[:, :n_e], g_x0[:, n_e:]
    else:
        g_xd, g_pd = g_x0, g_x0
    g['embed'] = g_xd / data['d_norm'].to(g_xd.dtype).view(-1, 1)
    feat = _nonidentity_features(data, p['embed'])
    if feat is not None:
        g['embed'] = torch.sparse.mm(feat.t(), g['embed'].contiguous())
    g_x_all, g['hgcn.weight'] = hier_conv_bwd(g_pd, saved['x_all_rows'], data['dp_edge_index'],
                                              p['hgcn.weight'], data['n_prot'], saved['h'])
    g_h2 = g_x_all[:data['n_prot']]
    (g['pp_encoder.conv1.lin.weight'], g['pp_encoder.conv1.bias'],
     g['pp_encoder.conv2.lin.weight'], g['pp_encoder.conv2.bias']) = pp_encoder_bwd(
        g_h2, p['pp_encoder.conv1.lin.weight'], p['pp_encoder.conv2.lin.weight'], saved['pp'])
    return g


# ---------------------------------------------------------------------------------------------
# A6 / A8  DistMult decoder and the loss  (src/layers.py:581-595, :338-340)
# ---------------------------------------------------------------------------------------------
def distmult_fwd(z, edge_index, edge_type, weight, sigmoid=True):
    v = (z[edge_index[0]] * z[edge_index[1]] * weight[edge_type]).sum(dim=1)
    return torch.sigmoid(v) if sigmoid else v


def distmult_bwd(g_score, z, edge_index, edge_type, weight, sigmoid=True):
    """Returns (g_z, g_weight) for upstream gradient g_score [M]."""
    zu, zv, dr = z[edge_index[0]], z[edge_index[1]], weight[edge_type]
    if sigmoid:
        s = torch.sigmoid((zu * zv * dr).sum(dim=1))
        g_score = g_score * s * (1 - s)
    gl = g_score.unsqueeze(1)
    g_z = torch.zeros_like(z)
    g_z.index_add_(0, edge_index[0], gl * zv * dr)
    g_z.index_add_(0, edge_index[1], gl * zu * dr)
    g_w = torch.zeros_like(weight)
    g_w.index_add_(0, edge_type, gl * zu * zv)
    return g_z, g_w


def nn_decoder_fwd(z, edge_index, edge_type, w1_l1, w1_l2, w2_l1, w2_l2):
    """NNDecoder.forward (src/layers.py:620-631), literal op order."""
    d1 = torch.relu(z[edge_index[0]] @ w1_l1)
    d2 = torch.relu(z[edge_index[1]] @ w2_l1)
    return torch.sigmoid((d1 * w1_l2[edge_type]).sum(dim=1) + (d2 * w2_l2[edge_type]).sum(dim=1))


def nn_decoder_bwd(g_score, z, edge_index, edge_type, w1_l1, w1_l2, w2_l1, w2_l2, chunk=1 << 20):
    """Explicit (autograd-free) backward of `nn_decoder_fwd` (src/layers.py:620-631 under autograd), in chunks of triples:
    -> (g_z, g_w1_l1, g_w1_l2, g_w2_l1, g_w2_l2) for the upstream gradient g_score of the sigmoid scores."""
    g_z = torch.zeros_like(z)
    g = [torch.zeros_like(w) for w in (w1_l1, w1_l2, w2_l1, w2_l2)]
    for b in range(0, edge_index.shape[1], chunk):
        u, v, et = edge_index[0, b:b + chunk], edge_index[1, b:b + chunk], edge_type[b:b + chunk]
        zu, zv = z[u], z[v]
        a1, a2 = zu @ w1_l1, zv @ w2_l1
        d1, d2 = torch.relu(a1), torch.relu(a2)
        s = torch.sigmoid((d1 * w1_l2[et]).sum(dim=1) + (d2 * w2_l2[et]).sum(dim=1))
        gx = (g_score[b:b + chunk] * s * (1 - s)).unsqueeze(1)
        g[1].index_add_(0, et, gx * d1)
        g[3].index_add_(0, et, gx * d2)
        ga1 = gx * w1_l2[et] * (a1 > 0)
        ga2 = gx * w2_l2[et] * (a2 > 0)
        g[0] += zu.t() @ ga1
        g[2] += zv.t() @ ga2
        g_z.index_add_(0, u, ga1 @ w1_l1.t())
        g_z.index_add_(0, v, ga2 @ w2_l1.t())
    return (g_z,) + tuple(g)


def tip_loss(pos_score, neg_score):
    """-mean log(pos + eps) - mean log(1 - neg + eps)   (src/layers.py:338-340)."""
    return -torch.log(pos_score + EPS).mean() - torch.log(1 - neg_score + EPS).mean()


def tip_loss_bwd(pos_score, neg_score):
    """d loss / d pos_score, d loss / d neg_score."""
    return (-1.0 / (pos_score + EPS) / pos_score.numel(),
            1.0 / (1 - neg_score + EPS) / neg_score.numel())


# ---------------------------------------------------------------------------------------------
# A7  typed negative sampling  (src/neg_sampling.py:5-26)
# ---------------------------------------------------------------------------------------------
def negative_sampling(pos_edge_index, num_nodes, rng=np.random):
    """Literal semantics of :5-19 including its resample quirk: the loop overwrites
    `perm[rest]` with fresh draws but recomputes `rest` as positions inside `tmp` (not `perm`),
    so later rounds re-index into the wrong array and a few positives survive."""
    idx = (pos_edge_index[0] * num_nodes + pos_edge_index[1]).numpy()
    perm = rng.choice(num_nodes ** 2, idx.size)
    rest = np.flatnonzero(np.isin(perm, idx))
    while rest.size > 0:
        tmp = rng.choice(num_nodes ** 2, rest.size)
        perm[rest] = tmp
        rest = np.flatnonzero(np.isin(tmp, idx))
    perm = torch.from_numpy(perm)
    return torch.stack([perm // num_nodes, perm % num_nodes]).long()


def typed_negative_sampling(pos_edge_index, num_nodes, range_list, rng=np.random):
    return torch.cat([negative_sampling(pos_edge_index[:, int(a):int(b)], num_nodes, rng)
                      for a, b in range_list], dim=1)


# ---------------------------------------------------------------------------------------------
# A10  per-relation metrics  (src/utils.py:86-93) -- sklearn, exactly as the reference calls it
# ---------------------------------------------------------------------------------------------
def auprc_auroc_ap(y, pred):
    from sklearn import metrics
    auroc, ap = metrics.roc_auc_score(y, pred), metrics.average_precision_score(y, pred)
    p, r, _ = metrics.precision_recall_curve(y, pred)
    return metrics.auc(r, p), auroc, ap


# ---------------------------------------------------------------------------------------------
# parameter initialisation by the reference's rules (for fixtures and the AUROC parity run)
# ---------------------------------------------------------------------------------------------
def init_params(n_drug, n_prot, n_rel, prot_drug_dim=16, n_embed=48, n_hid1=32, n_hid2=16,
                num_base=32, mod='cat', pp_hid1=32, pp_hid2=16, seed=1111, dtype=torch.float32):
    """Random parameters following the reference init rules (glorot for GCN `lin`, zeros bias,
    normal(0,1) embed :552, hgcn normal(1/sqrt(in)) :222, rgcn :142-152, decoder :595).  The
    draw ORDER is this file's own (torch RNG streams differ across versions anyway); parity
    runs always exchange explicit weights."""
    g = torch.Generator().manual_seed(seed)

    def normal(shape, std):
        return (torch.randn(shape, generator=g, dtype=torch.float64) * std).to(dtype)

    def glorot(o, i):
        a = math.sqrt(6.0 / (i + o))
        return ((torch.rand((o, i), generator=g, dtype=torch.float64) * 2 - 1) * a).to(dtype)

    d_in = n_embed + prot_drug_dim if mod == 'cat' else n_embed
    p = {'embed': normal((n_drug, n_embed), 1.0),
         'pp_encoder.conv1.lin.weight': glorot(pp_hid1, n_prot),
         'pp_encoder.conv1.bias': torch.zeros(pp_hid1, dtype=dtype),
         'pp_encoder.conv2.lin.weight': glorot(pp_hid2, pp_hid1),
         'pp_encoder.conv2.bias': torch.zeros(pp_hid2, dtype=dtype),
         'hgcn.weight': normal((pp_hid2, prot_drug_dim), 1 / math.sqrt(pp_hid2)),
         'rgcn1.basis': normal((num_base, d_in, n_hid1), 1 / math.sqrt(d_in)),
         'rgcn1.att': normal((n_rel, num_base), 1 / math.sqrt(num_base)),
         'rgcn1.root': normal((d_in, n_hid1), 1 / math.sqrt(d_in)),
         'rgcn2.basis': normal((num_base, n_hid1, n_hid2), 2 / n_hid1),
         'rgcn2.att': normal((n_rel, num_base), 1 / math.sqrt(num_base)),
         'rgcn2.root': normal((n_hid1, n_hid2), 2 / n_hid1),
         'decoder.weight': normal((n_rel, n_hid2), 1 / math.sqrt(n_hid2))}
    return p
