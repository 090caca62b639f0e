"""MI355X-native TIP modules behind the reference's `nn.Module` surface (`src/layers.py`).

Class names, constructor/forward signatures, parameter names, shapes and init rules follow the
reference so that its `state_dict`s load and `tip.py`'s loop runs unchanged (SURVEY.md section
8(b)); the bodies are new: every layer is a `torch.autograd.Function` from `tip_amd.ops` that
launches the HIP kernels of libtipk through the C ABI.  There is no PyG and no CPU path.

Static-graph caching: the reference's `GCNConv(cached=True)` (`src/layers.py:386-387`) keeps the
normalised P-P graph after the first call.  Here EVERY layer caches its gather plans, keyed by the
identity (storage pointer, shape, version counter) of the edge tensors it was given, because all
three TIP graphs are constant during training.  Pass a new tensor (or modify in place) to rebuild.
"""
import math
import os
import pickle

import numpy as np
import torch
from torch import nn
from torch.nn import Parameter as Param

from . import encoder, ops, switches
from .data import Data, build_data_dict
from .neg_sampling import typed_negative_sampling
from .plan import (build_pair_bwd_plan, build_dest_plan, build_row_stream_plan, build_row_stream_plan_s, build_gather_plan, build_gather_plan_segmented, build_rel_plan, build_stream_plan, build_stream_plan_rows, build_csr_plan, group_slots_for,
                   relations_per_segment, DEFAULT_CHUNK)
from .utils import process_edges, auprc_auroc_ap_by_range

EPS = 1e-13                    # src/layers.py:15

__all__ = ['GCNConv', 'MyRGCNConv', 'MyRGCNConv2', 'MyHierarchyConv', 'PPEncoder', 'FMEncoder',
           'FMEncoderCat', 'MultiInnerProductDecoder', 'NNDecoder', 'Setting', 'TIP']


# ---------------------------------------------------------------------------------------------
# plan caches
# ---------------------------------------------------------------------------------------------
def _ident(*tensors):
    key = []
    for t in tensors:
        if t is None:
            key.append(None)
        elif t.is_sparse:                        # no storage pointer: the (pinned) object itself
            key.append((id(t), tuple(t.shape), str(t.device), t.dtype))
        else:
            key.append((t.data_ptr(), tuple(t.shape), t._version, str(t.device), t.dtype))
    return tuple(key)


class _PlanCache(object):
    """One slot per layer (graphs are static; a changed tensor identity rebuilds)."""

    def __init__(self):
        self.key, self.value, self.pins = None, None, None

    def get(self, key_tensors, builder):
        key = _ident(*key_tensors)
        if key != self.key:
            self.value = builder()
            self.key, self.pins = key, key_tensors          # keep the tensors alive: pointers stay unique
        return self.value

    # A cache is derived state (device plans, lazily built through closures): it is never pickled.
    # `torch.save(model, ...)` as in the reference's tip.py:36, `copy.deepcopy` and multiprocessing
    # therefore work after a forward pass; the plans are rebuilt on the first call after loading.
    def __getstate__(self):
        return {}

    def __setstate__(self, state):
        self.key, self.value, self.pins = None, None, None

    def __deepcopy__(self, memo):
        return _PlanCache()


def _is_identity_features(x):
    """True for the `sparse_id(n)` features of prepare.py:22-23 (cached on the tensor object)."""
    if x is None:
        return True
    flag = getattr(x, '_tipk_identity', None)
    if flag is None:
        flag = False
        if x.is_sparse and x.shape[0] == x.shape[1]:
            xc = x.coalesce()
            idx, val = xc.indices(), xc.values()
            n = x.shape[0]
            ar = torch.arange(n, device=idx.device)
            flag = bool(idx.shape[1] == n and (idx[0] == ar).all() and (idx[1] == ar).all() and (val == 1).all())
        try:
            x._tipk_identity = flag
        except Exception:
            pass
    return flag


def _sparse_feature_graph(x, chunk):
    """Plans for y = x_sparse @ table (general sparse features, e.g. mono side-effect columns)."""
    xc = x.coalesce()
    r, c, v = xc.indices()[0], xc.indices()[1], xc.values().to(torch.float32)
    n, m = x.shape
    return ops.AggGraph(build_gather_plan(r, c, n, m, v, chunk), build_gather_plan(c, r, m, n, v, chunk))


class _FeatureInput(nn.Module):
    """`x @ W` for identity / sparse / dense `x` -- the three feature kinds the reference's
    `torch.matmul(x_drug, embed)` (:532) and `GCNConv.lin` can meet."""

    def __init__(self):
        super().__init__()
        self._cache = _PlanCache()

    def apply_table(self, x, table):
        """table: [in, out] (already in x @ table orientation)."""
        if _is_identity_features(x):
            return table
        if x.is_sparse:
            graph = self._cache.get((x,), lambda: _sparse_feature_graph(x, DEFAULT_CHUNK))
            return ops.aggregate(table, graph)
        return ops.matmul(x, table)


# ---------------------------------------------------------------------------------------------
# A1  GCNConv / PPEncoder   (src/layers.py:380-395; GCNConv = PyG 2.0.1 semantics)
# ---------------------------------------------------------------------------------------------
class _Lin(nn.Module):
    """bias-free linear map with glorot weight [out, in] (PyG `Linear(weight_initializer='glorot')`)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        # shape [out, in] as in PyG (state_dicts load unchanged), MEMORY [in, out]: with identity features
        # lin(I) = W^T is then the parameter's own storage and d W = (d lin)^T a view of the gradient the
        # aggregation kernel wrote -- no transpose kernels on either pass (19 081 x 32 at BioSNAP)
        self.weight = Param(torch.empty(in_channels, out_channels).t())
        bound = math.sqrt(6.0 / (in_channels + out_channels))
        self.weight.data.uniform_(-bound, bound)


def gcn_norm_graph(edge_index, num_nodes, chunk=DEFAULT_CHUNK, d=None, rows=None):
    """Normalised adjacency D^-1/2 (A + I) D^-1/2 as a pair of gather plans: existing self loops
    are replaced by exactly one unit loop per node, deg = in-degree incl. the loop, inf -> 0.
    d: width of the rows the plans will aggregate (enables in-workgroup combination of split rows).
    rows: ascending int64 node ids -- only THESE output rows are produced, as a compact [len(rows), d] matrix (same
    weights, same edge order inside a row as in the full graph; the transposed plan reads the compact gradient and
    writes all num_nodes rows).  FMEncoder uses it for conv2: only the proteins that are P -> D sources are ever read."""
    G = group_slots_for(d) if d else 0
    row, col = edge_index[0].to(torch.int64), edge_index[1].to(torch.int64)
    keep = row != col
    loop = torch.arange(num_nodes, device=row.device)
    row, col = torch.cat([row[keep], loop]), torch.cat([col[keep], loop])
    deg = torch.bincount(col, minlength=num_nodes).to(torch.float32)
    dis = deg.pow(-0.5)
    dis[torch.isinf(dis)] = 0
    w = dis[row] * dis[col]
    if rows is not None:
        rows = rows.to(torch.int64).to(row.device)
        inv = torch.full((num_nodes,), -1, dtype=torch.int64, device=row.device)
        inv[rows] = torch.arange(rows.numel(), device=row.device)
        sel = inv[col] >= 0
        row_s, col_s, w_s = row[sel], inv[col[sel]], w[sel]
        n_s = int(rows.numel())
        return ops.AggGraph(build_gather_plan(col_s, row_s, n_s, num_nodes, w_s, chunk, 'pp.fwd.rows', G),
                            build_gather_plan(row_s, col_s, num_nodes, n_s, w_s, chunk, 'pp.bwd.rows', G))
    return ops.AggGraph(build_gather_plan(col, row, num_nodes, num_nodes, w, chunk, 'pp.fwd', G),
                        build_gather_plan(row, col, num_nodes, num_nodes, w, chunk, 'pp.bwd', G))


class GCNConv(nn.Module):
    """out = A_hat (x W^T) + bias.  `cached` is accepted for signature parity; plans are always
    cached by edge-tensor identity."""

    def __init__(self, in_channels, out_channels, cached=False, bias=True, chunk=DEFAULT_CHUNK):
        super().__init__()
        self.in_channels, self.out_channels, self.cached, self.chunk = in_channels, out_channels, cached, chunk
        self.lin = _Lin(in_channels, out_channels)
        if bias:
            self.bias = Param(torch.zeros(out_channels))
        else:
            self.register_parameter('bias', None)
        self._feat = _FeatureInput()
        self._cache = _PlanCache()
        self._cache_rows = _PlanCache()

    def aggregates_first(self, x, rows):
        """True if forward(x, ..., rows=rows) takes the aggregate-first route (the only one that honours gate_input)."""
        return rows is not None and not x.is_sparse and not _is_identity_features(x) and \
            ops.gather_sum_lin_supported(self.in_channels, self.out_channels, group_slots_for(self.in_channels))

    def forward(self, x, edge_index, fuse_relu=False, rows=None, gate_input=False, link=None):
        """rows (extension): ascending int64 node ids -- return only these rows of the layer's output, [len(rows), out]
        (`gcn_norm_graph(rows=...)`: the rows nobody reads are neither aggregated nor back-propagated through).
        fuse_relu='gated_downstream' / gate_input / link (extension, `ops.GateLink`): the ReLU hand-over between this layer
        and the ONE consumer of its output (FMEncoder: conv1 -> conv2)."""
        n = x.shape[0]
        if gate_input and not self.aggregates_first(x, rows):
            raise NotImplementedError('gate_input is wired for the aggregate-first route only')
        if self.aggregates_first(x, rows):
            # few rows kept: aggregate first, the dense map on the kept rows in the gather's own launch (ops._GCNConvAggFirst);
            # the plans move rows of in_channels floats
            graph = self._cache_rows.get((edge_index, rows),
                                         lambda: gcn_norm_graph(edge_index, n, self.chunk, self.in_channels, rows))
            return ops.gcn_conv_agg_first(x, self.lin.weight, self.bias, graph, fuse_relu, gate_input, link)
        if rows is not None:
            graph = self._cache_rows.get((edge_index, rows), lambda: gcn_norm_graph(edge_index, n, self.chunk, self.out_channels, rows))
        else:
            graph = self._cache.get((edge_index,), lambda: gcn_norm_graph(edge_index, n, self.chunk, self.out_channels))
        if _is_identity_features(x):
            return ops.gcn_conv(None, self.lin.weight, self.bias, graph, fuse_relu, link)      # lin(I) = W^T
        if x.is_sparse:
            xl = self._feat.apply_table(x, ops.linear_t(None, self.lin.weight))
            return ops.aggregate(xl, graph, bias=self.bias, relu=fuse_relu)
        return ops.gcn_conv(x, self.lin.weight, self.bias, graph, fuse_relu, link)

    def __repr__(self):
        return 'GCNConv(%d, %d)' % (self.in_channels, self.out_channels)


class PPEncoder(nn.Module):
    def __init__(self, in_dim, hid1=32, hid2=16):
        super().__init__()
        self.out_dim = hid2
        self.conv1 = GCNConv(in_dim, hid1, cached=True)
        self.conv2 = GCNConv(hid1, hid2, cached=True)

    def forward(self, x, edge_index):
        x = self.conv1(x, edge_index, fuse_relu=True)                # ReLU fused into the epilogue
        return self.conv2(x, edge_index)


# ---------------------------------------------------------------------------------------------
# A2  MyHierarchyConv   (src/layers.py:196-247)
# ---------------------------------------------------------------------------------------------
def hier_graph(edge_index, n_all, n_source, chunk=DEFAULT_CHUNK, table_rows=None, d=None):
    """mean over incoming edges in the concatenated node space, rows [n_source:] only.
    table_rows: rows of the table actually handed to the kernel (n_all for the concatenated
    tensor, n_source when only the source block is passed and no edge starts beyond it)."""
    src, dst = edge_index[0].to(torch.int64), edge_index[1].to(torch.int64)
    keep = dst >= n_source
    src, dst = src[keep], dst[keep] - n_source
    n_t = n_all - n_source
    n_tab = n_all if table_rows is None else table_rows
    cnt = torch.bincount(dst, minlength=n_t).to(torch.float32).clamp_(min=1)
    G = group_slots_for(d) if d else 0
    scale = (1.0 / cnt).contiguous()
    # the transposed plan carries 1/count as per-edge weights: no scaling pass before the backward gather
    graph = ops.AggGraph(build_gather_plan(dst, src, n_t, n_tab, None, chunk, 'pd.fwd', G),
                         build_gather_plan(src, dst, n_tab, n_t, scale[dst], chunk, 'pd.bwd', G), scale,
                         bwd_scaled=True)
    # CSR by target (edge order kept inside a target) for the fused P -> D + drug-mix forward launch (ops.drug_mix_gather;
    # include/tipk.h section 3)
    i32 = lambda t: t.to(torch.int32).contiguous()
    of = torch.sort(dst, stable=True).indices
    ptr = lambda idx, n: i32(torch.cat([idx.new_zeros(1), torch.cumsum(torch.bincount(idx, minlength=n), 0)]))
    # workgroups of the forward launch (16 wavefronts): targets dealt by edge count -- one with more than 512 edges has a
    # workgroup to itself (W = 16 wavefronts on it: one BioSNAP drug has 2 834 protein targets), those with 65 ... 512 edges go
    # four to a workgroup (W = 4), the others sixteen (a wavefront each); desc = {first index into `order`, n | W << 8}
    order, wgs = drug_workgroups(torch.bincount(dst, minlength=n_t))
    fwd_wg = torch.tensor(wgs, dtype=torch.int32, device=src.device).view(-1, 2).contiguous()
    graph.pd_csr = dict(fwd_ptr=ptr(dst, n_t), fwd_src=i32(src[of]), scale=scale, fwd_wg=fwd_wg, fwd_order=i32(order), n_src=int(n_tab))
    # CSR by SOURCE row (edge order kept inside a row) with 1 / count of the edge's target: the transposed gather inside the
    # fused backward launch of the stage (tipk_pd_stage_bwd, tip_amd/encoder.py)
    ot = torch.sort(src, stable=True).indices
    graph.pd_csr.update(t_ptr=ptr(src, n_tab), t_dst=i32(dst[ot]), t_w=scale[dst[ot]].contiguous())
    # ... and the deal of the source rows to that launch's workgroups: consecutive rows, at most max_rows of them and at most
    # max_edges edges (a row with more edges than that has a workgroup to itself)
    if src.is_cuda:
        import ctypes
        mr, me = ctypes.c_int(0), ctypes.c_int(0)
        ops.lib().tipk_pd_stage_bwd_limits(ctypes.byref(mr), ctypes.byref(me))
        bounds = deal_rows_by_edges(torch.bincount(src, minlength=n_tab).tolist(), mr.value, me.value)
        graph.pd_csr['t_wg'] = torch.tensor(bounds, dtype=torch.int32, device=src.device)
    return graph


def deal_rows_by_edges(counts, max_rows, max_edges):
    """Boundaries [0, ..., len(counts)] of consecutive row blocks with at most max_rows rows and -- unless a single row alone has
    more -- at most max_edges edges each (the row workgroups of tipk_pd_stage_bwd).  BioSNAP's P -> D graph has ~100 consecutive
    proteins that 65 ... 94 drugs target: blocks of 64 ROWS gave three workgroups 2 200 ... 3 300 edges against a mean of 326."""
    bounds, rows_in, edges_in = [0], 0, 0
    for r_, c_ in enumerate(counts):
        if rows_in and (rows_in == max_rows or edges_in + c_ > max_edges):
            bounds.append(r_)
            rows_in, edges_in = 0, 0
        rows_in += 1
        edges_in += c_
    bounds.append(len(counts))
    return bounds


def drug_workgroups(counts):
    """(order, descs) of the forward P -> D launch: rows by decreasing edge count; desc = [first index into order, n | W << 8] --
    a row with more than 512 edges alone on the 16 wavefronts of a workgroup (W = 16), rows with 65 ... 512 edges four to a
    workgroup (W = 4), the others sixteen (W = 1).  Every row exactly once."""
    cnt_t = torch.as_tensor(counts)
    n_t = int(cnt_t.numel())
    order = torch.sort(cnt_t, descending=True, stable=True).indices
    cs = cnt_t[order].tolist()
    n_big = sum(1 for c_ in cs if c_ > 512)
    n_mid = sum(1 for c_ in cs if 64 < c_ <= 512)
    wgs = [[i, 1 | (16 << 8)] for i in range(n_big)]
    wgs += [[b0, min(4, n_big + n_mid - b0) | (4 << 8)] for b0 in range(n_big, n_big + n_mid, 4)]
    wgs += [[b0, min(16, n_t - b0) | (1 << 8)] for b0 in range(n_big + n_mid, n_t, 16)]
    return order, wgs


class MyHierarchyConv(nn.Module):
    """Directed protein -> drug mean aggregation followed by a dense map."""

    def __init__(self, in_dim, out_dim, unigue_source_num, unique_target_num,
                 is_after_relu=True, is_bias=False, chunk=DEFAULT_CHUNK):
        super().__init__()
        self.in_dim, self.out_dim = in_dim, out_dim
        self.unique_source_num, self.unique_target_num = unigue_source_num, unique_target_num
        self.is_after_relu, self.chunk = is_after_relu, chunk
        self.weight = Param(torch.empty(in_dim, out_dim))
        if is_bias:
            # the reference's bias branch cannot run (`if self.bias:` on a tensor, and an
            # out_dim-sized bias added to in_dim-wide rows, :236-237); every caller passes False
            raise NotImplementedError('MyHierarchyConv(is_bias=True) is not executable in the reference either')
        self.register_parameter('bias', None)
        self._cache = _PlanCache()
        self._cache_src = _PlanCache()
        self._cache_src_rows = _PlanCache()
        self._cache_rows = _PlanCache()
        self.reset_parameters()

    def reset_parameters(self):
        scale = 1.0 if self.is_after_relu else 2.0
        self.weight.data.normal_(std=scale / np.sqrt(self.in_dim))

    def forward(self, x, edge_index, range_list=None):
        n_all = x.shape[0]
        graph = self._cache.get((edge_index,), lambda: hier_graph(edge_index, n_all, self.unique_source_num,
                                                                  self.chunk, d=self.in_dim))
        mean = ops.aggregate(x, graph)
        out = ops.matmul(mean, self.weight)
        assert out.shape[0] == self.unique_target_num
        return out

    def forward_sources(self, x_src, edge_index):
        """Same result as `forward(cat(x_src, zeros[n_target]), edge_index, ...)` without building
        the concatenation (the reference's zero rows `hdrug`, :526, are never read when every edge
        starts at a source node).  Returns None if some edge starts beyond the source block."""
        n_src = self.unique_source_num
        n_all = n_src + self.unique_target_num

        def build():
            if edge_index.numel() and int(edge_index[0].max()) >= n_src:
                return None
            return hier_graph(edge_index, n_all, n_src, self.chunk, table_rows=n_src, d=self.in_dim)
        graph = self._cache_src.get((edge_index,), build)
        if graph is None:
            return None
        return ops.matmul(ops.aggregate(x_src, graph), self.weight)

    def source_rows(self, edge_index):
        """Ascending ids of the source nodes some edge starts at (the only rows of the source block the mean reads), or
        None if an edge starts beyond the source block.  Cached per edge tensor."""
        n_src = self.unique_source_num

        def build():
            if edge_index.numel() and int(edge_index[0].max()) >= n_src:
                return None
            return torch.unique(edge_index[0].to(torch.int64))
        return self._cache_rows.get((edge_index,), build)

    def mean_sources(self, x_src, edge_index, rows=None, graph_only=False):
        """The mean aggregate of `forward_sources` WITHOUT the dense map (the caller applies
        `self.weight` fused with what follows, ops.drug_mix_mm); None if unusable.
        rows: `source_rows(edge_index)` -- x_src then holds only those rows of the source block, in that order."""
        n_src = self.unique_source_num
        n_all = n_src + self.unique_target_num

        def build():
            if edge_index.numel() and int(edge_index[0].max()) >= n_src:
                return None
            if rows is None:
                return hier_graph(edge_index, n_all, n_src, self.chunk, table_rows=n_src, d=self.in_dim)
            inv = torch.full((n_src,), -1, dtype=torch.int64, device=edge_index.device)
            inv[rows] = torch.arange(rows.numel(), device=edge_index.device)
            src = inv[edge_index[0].to(torch.int64)]
            assert bool((src >= 0).all())
            n_c = int(rows.numel())                                        # compact source block + the targets behind it
            ei = torch.stack([src, edge_index[1].to(torch.int64) - n_src + n_c])
            return hier_graph(ei, n_c + self.unique_target_num, n_c, self.chunk, table_rows=n_c, d=self.in_dim)
        graph = (self._cache_src if rows is None else self._cache_src_rows).get((edge_index, rows), build)
        if graph is None:
            return None
        if graph_only:
            return graph
        return ops.aggregate(x_src, graph)

    def __repr__(self):
        return 'MyHierarchyConv(%d, %d)' % (self.in_dim, self.out_dim)


# ---------------------------------------------------------------------------------------------
# A4/A5  R-GCN layers   (src/layers.py:21-99, :102-193)
# ---------------------------------------------------------------------------------------------
def relation_of_edges(range_list, n_edges, device):
    rg = torch.as_tensor(range_list).to(torch.int64).cpu()
    if rg.numel() == 0:                                       # a rank without relations (more ranks than relations)
        if n_edges:
            raise ValueError('empty range_list for %d edges' % n_edges)
        return torch.zeros(0, dtype=torch.int64, device=device)
    sizes = rg[:, 1] - rg[:, 0]
    if not (int(sizes.sum()) == n_edges and int(rg[0, 0]) == 0 and bool((rg[1:, 0] == rg[:-1, 1]).all())):
        raise ValueError('range_list must be consecutive blocks covering all %d edges' % n_edges)
    return torch.repeat_interleave(torch.arange(rg.shape[0]), sizes).to(device)


def _shared_pair_bwd_plan(src, dst, rel, n_nodes, n_rel, scale, symmetric, n_wg, lanes, share=None):
    """`build_pair_bwd_plan`; the R-GCN layers of one model run on the same graph and share the plan (it does not depend on the
    layer's width).  share = (dict owned by the model, (edge_index, range_list) as the caller passed them): the entry keeps
    those tensors alive, so their identity + version IS the edge list -- no checksums, no device sync, nothing global."""
    build = lambda: build_pair_bwd_plan(src, dst, rel, n_nodes, n_rel, scale, symmetric, n_wg, lanes, ops.rel_stream_piece())
    if share is None:
        return build()
    cache, owners = share
    key = tuple((id(t), t._version) if torch.is_tensor(t) else None for t in owners) + \
        (int(n_nodes), int(n_rel), bool(symmetric), int(n_wg), int(lanes))
    hit = cache.get(key)
    if hit is None:
        while len(cache) >= 2:                                # (a training and an evaluation graph)
            cache.pop(next(iter(cache)))
        hit = cache[key] = (build(), owners)
    return hit[0]


def pair_link_words(src, dst, n_nodes):
    """uint32 words (as int32) [n_nodes padded to 8][ceil(n_nodes / 32)]: bit r of word (u, t) = some edge links u -> 32 t + r."""
    n8, n32 = -(-n_nodes // 8) * 8, -(-n_nodes // 32) * 32
    link = torch.zeros((n8, n32), dtype=torch.bool, device=src.device)
    link[src, dst] = True
    w = (link.view(n8, n32 // 32, 32).to(torch.int64) << torch.arange(32, device=src.device).view(1, 1, 32)).sum(-1)
    return torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32).contiguous()


def rgcn_graph(edge_index, rel, n_nodes, n_rel, chunk=DEFAULT_CHUNK, in_degree=None, d_out=None, n_bases=None, paired=False,
               plan_share=None):
    """fwd: destination <- row (rel*N + src) of Y;  bwd: (rel*N + src) <- destination row of g;
    scale = 1 / max(1, in-degree over ALL relations) (torch-scatter 'mean').  `in_degree`: [N]
    in-degree of the WHOLE graph when `edge_index` is only one rank's shard (tip_amd/dist.py)."""
    src, dst = edge_index[0].to(torch.int64), edge_index[1].to(torch.int64)
    rel = rel.to(torch.int64)
    if rel.numel() and (int(rel.min()) < 0 or int(rel.max()) >= n_rel):
        raise IndexError('edge_type out of range')
    yrow = rel * n_nodes + src
    if in_degree is None:
        deg = torch.bincount(dst, minlength=n_nodes).to(torch.float32).clamp_(min=1)
    else:
        assert in_degree.numel() == n_nodes
        deg = in_degree.to(src.device).to(torch.float32).clamp_(min=1)
    rl_fwd = rl_bwd = rs_bwd = pair_fwd = pair_bwd = None
    if n_nodes <= 1024 and n_rel > 0 and src.numel() > 0:
        # relation-local plans for the LDS-resident kernels (used when a relation's table fits in LDS)
        n_cu = torch.cuda.get_device_properties(src.device).multi_processor_count if src.is_cuda else 256
        on_dev = bool(d_out and src.is_cuda)
        # workgroups of the pair-form gathers PER LAYER.  paired: the layer is one of the two of an FMEncoder, which gathers both
        # layers' cells, and both layers' d att, in one launch each (tip_amd/encoder.py) -- 2 x (CUs / 2) workgroups are one round
        # of the chip (140 KB of LDS: one workgroup per CU); with CUs per layer every CU staged its table twice (25.2 -> 21.1 us
        # and 28.0 -> 24.4 us at BioSNAP).  A layer on its own fills the chip by itself.  TIPK_PAIR_WGS: experiments
        pair_wgs = int(os.environ.get('TIPK_PAIR_WGS', 0)) or (max(1, n_cu // 2) if paired else n_cu)
        # launch = (workgroups x column blocks), sized for 1 or 2 workgroups per CU (tipk_rel_gather_occupancy)
        wg_f = (ops.rel_gather_wgs(n_nodes, d_out, False, n_cu) if on_dev else 0) or n_cu
        wg_b = (ops.rel_gather_wgs(n_nodes, d_out, True, n_cu) if on_dev else 0) or n_cu
        # lanes per slot of the launch (columns of one column block / 4): the plans order every run for
        # conflict-free LDS reads of that shape and pre-scale the ids to row offsets
        split_f = ops.rel_gather_split(n_nodes, d_out, False) if on_dev else 0
        split_b = ops.rel_gather_split(n_nodes, d_out, True) if on_dev else 0
        lanes_f = (d_out // split_f) // 4 if split_f else None
        lanes_b = (d_out // split_b) // 4 if split_b else None
        cap_f = None           # (cutting forward units to the id chunk was measured slower: smaller units fill fewer bands)
        rl_fwd = lambda: build_rel_plan(dst, src, rel, n_nodes, n_rel, wg_f, lanes=lanes_f, unit_cap=cap_f)
        # forward pass in pair form: cells (source, destination) <- sums of att rows (LDS-resident att table)
        split_p = ops.stream_gather_split(n_rel, n_bases) if on_dev and n_bases and n_nodes * n_nodes < 2 ** 24 else 0
        if split_p:
            # every relation symmetric (u -> v iff v -> u: BioSNAP) => cell (u, v) == cell (v, u): build the cells
            # with u <= v only, from half the edges; the product reads the mirrored ones at their transposed place
            k_fw = torch.sort((rel * n_nodes + src) * n_nodes + dst).values
            k_bw = torch.sort((rel * n_nodes + dst) * n_nodes + src).values
            symmetric = bool(torch.equal(k_fw, k_bw)) and bool(ops.lib().tipk_pair_product_supported(n_bases, d_out))
            keep = src <= dst if symmetric else torch.ones_like(src, dtype=torch.bool)
            pair_fwd = build_stream_plan_rows(src[keep] * n_nodes + dst[keep], rel[keep], n_nodes * n_nodes, n_rel, pair_wgs,
                                              (n_bases // split_p) // 4, ops.rel_stream_piece())
            pair_fwd.symmetric = symmetric
            # which pairs are linked at all, one bit per (source, destination): the product fetches only their cells
            # (uint32 [sources padded to 8][ceil(N / 32)], bit r of word (u, t) = pair (u, 32 t + r); tipk.h section 2c `links`)
            pair_fwd.links = pair_link_words(src, dst, n_nodes)
        # transposed pass: wave streams (no work units, no barriers) when g' fits in LDS, else relation-local units
        split_s = ops.rel_stream_split(n_nodes, d_out) if on_dev and n_rel * n_nodes < 2 ** 24 else 0
        if pair_fwd is not None and ops.pair_grads_supported(n_bases, d_out):
            # backward pass in pair form as well (round 5): the cells of the forward pass + one 128-byte gradient row per
            # linked pair; d att walks half the edges of a symmetric graph.  The plan does not depend on the layer's width:
            # the layers of a model share it
            sym, lanes_p = pair_fwd.symmetric, (n_bases // split_p) // 4
            pair_bwd = lambda: _shared_pair_bwd_plan(src, dst, rel, n_nodes, n_rel, 1.0 / deg, sym, pair_wgs, lanes_p, plan_share)
        if split_s:
            # compact node-major rows when the products of dY can run on them (tipk_rgcn_node_products); with the pair-form
            # backward pass the plan is only built if some pass asks for it (a forward pass on the Y route of a sharded run)
            compact = bool(n_bases) and ops.node_products_slabs(n_nodes, d_out, n_rel, n_bases) > 0
            lanes_s = (d_out // split_s) // 4
            build_rs = lambda: build_stream_plan(src, dst, rel, n_nodes, n_rel, n_cu, lanes_s, ops.rel_stream_piece(), compact=compact)
            rs_bwd = build_rs if pair_bwd is not None else build_rs()
        else:
            rl_bwd = lambda: build_rel_plan(src, dst, rel, n_nodes, n_rel, wg_b, backward=True, lanes=lanes_b)
    # forward pass of LARGE node sets without Y (tipk_rgcn_dest_products): where Y = [R N, d_out] would not stay in the
    # Infinity Cache (config 5: 10 GB) and neither LDS route applies
    dest_fwd = row_fwd = row_bwd = None
    wide = False
    if src.is_cuda and d_out and n_bases and pair_fwd is None and n_rel * n_nodes * d_out * 4 > (192 << 20):
        bits = ops.dest_products_bits(n_nodes, n_rel, n_bases, 1)
        if bits:
            dest_fwd = lambda: build_dest_plan(src, dst, rel, n_nodes, n_rel, bits)
        # row sums assembled in LDS (tipk_rgcn_row_products): rows (relation, destination) <- sources for the forward pass,
        # rows (relation, source) <- destinations for the transposed pass.  Widths that are multiples of 64 take the
        # wave-uniform form (tipk_rgcn_row_products_s: any node count whose table stays below 2 GB); other multiples of 32 the
        # per-lane form (<= 65 536 nodes: a 16-bit node field in the entry word).  A plan is installed only where the
        # matching `_supported` query passes for THIS node count: the passes test `graph.row_fwd is not None`
        wide = bool(d_out) and d_out % 64 == 0 and ops.row_products_s_supported(n_nodes, n_rel, n_bases, d_out) \
            and n_nodes < (1 << 23) and n_nodes * d_out * 4 < 2 ** 31
        if wide:
            row_fwd = lambda: build_row_stream_plan_s(dst, src, rel, n_nodes, n_rel)
            row_bwd = lambda: build_row_stream_plan_s(src, dst, rel, n_nodes, n_rel)
        elif ops.row_products_supported(n_nodes, n_rel, n_bases, 32):
            row_fwd = lambda: build_row_stream_plan(dst, src, rel, n_nodes, n_rel)
            row_bwd = lambda: build_row_stream_plan(src, dst, rel, n_nodes, n_rel)

    def fwd_plan():
        # Y = [R N, d_out] beyond the Infinity Cache (config 5: 10 GB): launch the items relation block by
        # relation block, so that a row of Y gathered by several edges crosses the fabric once
        y_bytes = n_rel * n_nodes * (d_out or 0) * 4
        ordered = rel.numel() < 2 or bool((rel[1:] >= rel[:-1]).all())
        if y_bytes > (192 << 20) and ordered:
            per = relations_per_segment(n_nodes, d_out)
            return build_gather_plan_segmented(dst, yrow, rel // per, n_nodes, n_rel * n_nodes, chunk, 'dd.fwd')
        return build_gather_plan(dst, yrow, n_nodes, n_rel * n_nodes, None, chunk, 'dd.fwd')

    # the generic plans are built on first use: with the relation-local kernels they are never needed
    return ops.AggGraph(fwd_plan,
                        lambda: build_gather_plan(yrow, dst, n_rel * n_nodes, n_nodes, None, chunk, 'dd.bwd'),
                        (1.0 / deg).contiguous(), rl_fwd, rl_bwd,
                        csr_bwd=lambda: build_csr_plan(yrow, dst, n_rel * n_nodes, n_nodes, 'dd.bwd'), rs_bwd=rs_bwd,
                        pair_fwd=pair_fwd, pair_bwd=pair_bwd, dest_fwd=dest_fwd, row_fwd=row_fwd, row_bwd=row_bwd,
                        row_wave_uniform=wide)


class _RGCNBase(nn.Module):
    def __init__(self, in_channels, out_channels, num_relations, num_bases, after_relu, bias=False,
                 chunk=DEFAULT_CHUNK):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.num_relations, self.num_bases, self.after_relu, self.chunk = num_relations, num_bases, after_relu, chunk
        self.basis = Param(torch.empty(num_bases, in_channels, out_channels))
        self.att = Param(torch.empty(num_relations, num_bases))
        self.root = Param(torch.empty(in_channels, out_channels))
        if bias:
            self.bias = Param(torch.empty(out_channels))
        else:
            self.register_parameter('bias', None)
        self._cache = _PlanCache()
        self.shard = None                       # tip_amd.dist.RelationShard for multi-GPU runs
        self.paired = False                     # one of the two layers of an FMEncoder (sizes the pair plans: rgcn_graph)
        self.plan_share = None                  # dict shared with the other layers on the same graph (pair backward plan)
        self.reset_parameters()

    def reset_parameters(self):
        self.att.data.normal_(std=1 / np.sqrt(self.num_bases))
        std = 2 / self.in_channels if self.after_relu else 1 / np.sqrt(self.in_channels)
        self.root.data.normal_(std=std)
        self.basis.data.normal_(std=std)
        if self.bias is not None:
            self.bias.data.zero_()

    def _global_degree(self):
        """in-degree over the relations of ALL ranks when this layer holds one shard of them."""
        return None if self.shard is None else self.shard.in_degree

    def _run(self, x, graph, fuse_relu=False, gate_input=False, defer_output=False, partner=None, cells_token=None):
        if self.bias is not None:
            out = ops.rgcn(x, self.basis, self.att, self.root, graph, self.shard) + self.bias
            if gate_input or fuse_relu == 'gated_downstream':
                raise NotImplementedError('ReLU-mask hand-over between layers is only wired for bias=False')
            return torch.relu(out) if fuse_relu else out
        return ops.rgcn(x, self.basis, self.att, self.root, graph, self.shard, relu=fuse_relu, gate_input=gate_input,
                        defer_output=defer_output, partner=partner, cells_token=cells_token)

    def __repr__(self):
        return '%s(%d, %d, num_relations=%d)' % (self.__class__.__name__, self.in_channels, self.out_channels,
                                                 self.num_relations)


class MyRGCNConv2(_RGCNBase):
    """Range-list variant (:102-193): relation r owns edges `range_list[r] = (start, end)`;
    `edge_type` is accepted and ignored, as in the reference."""

    def graph_for(self, n, edge_index, range_list):
        """The layer's cached plans of this D-D graph over n nodes (built on first use)."""
        def build():
            rel = relation_of_edges(range_list, edge_index.shape[1], edge_index.device)
            return rgcn_graph(edge_index, rel, n, self.num_relations, self.chunk, in_degree=self._global_degree(),
                              d_out=self.out_channels, n_bases=self.num_bases, paired=self.paired and self.shard is None,
                              plan_share=None if self.plan_share is None else (self.plan_share, (edge_index, range_list)))
        return self._cache.get((edge_index, range_list if torch.is_tensor(range_list) else None), build)

    def forward(self, x, edge_index, edge_type, range_list, fuse_relu=False, gate_input=False, defer_output=False, next_layer=None,
                cells_token=None):
        """`fuse_relu` (extension): apply the ReLU that follows the layer in FMEncoder
        (src/layers.py:547) inside the layer's last kernel; 'gated_downstream' additionally leaves
        the ReLU's backward mask to the one consumer, which is called with `gate_input=True`.
        `defer_output` (extension, only with 'gated_downstream'): the layer's final slab sum may be left to that consumer --
        an R-GCN layer of this package, which runs it in the launch of its own XB product; the returned tensor must not be
        read by anything else.
        `next_layer` (extension): the R-GCN layer that follows on the SAME graph -- its pair cells (they depend on its `att`
        alone) are gathered in this layer's cell launch."""
        n = x.shape[0]
        graph = self.graph_for(n, edge_index, range_list)
        partner = None
        if next_layer is not None and self.shard is None and next_layer.shard is None and next_layer.bias is None and \
                next_layer.num_relations == self.num_relations and next_layer.num_bases == self.num_bases:
            partner = (next_layer.att, next_layer.graph_for(n, edge_index, range_list), next_layer.out_channels, cells_token)
        return self._run(x, graph, fuse_relu, gate_input, defer_output, partner, cells_token if next_layer is None else None)


class MyRGCNConv(_RGCNBase):
    """Per-edge-type variant (:21-99): same arithmetic, relation id taken from `edge_type`; the
    reference gathers an E x in x out weight tensor here, this build needs no edge ordering."""

    def forward(self, x, edge_index, edge_type):
        n = x.shape[0]
        graph = self._cache.get((edge_index, edge_type),
                                lambda: rgcn_graph(edge_index, edge_type, n, self.num_relations, self.chunk,
                                                   in_degree=self._global_degree(), d_out=self.out_channels,
                                                   n_bases=self.num_bases))
        return self._run(x, graph)


# ---------------------------------------------------------------------------------------------
# A3 + composition  FMEncoder   (src/layers.py:471-553; FMEncoderCat :401-468 is its cat mode)
# ---------------------------------------------------------------------------------------------
class FMEncoder(nn.Module):
    def __init__(self, device, in_dim_drug, num_dd_et, in_dim_prot, uni_num_prot, uni_num_drug,
                 prot_drug_dim=64, num_base=32, n_embed=64, n_hid1=32, n_hid2=16, mod='cat'):
        super().__init__()
        assert mod in {'add', 'cat'}
        if mod == 'add':
            assert n_embed == prot_drug_dim
        self.num_et, self.out_dim, self.mod = num_dd_et, n_hid2, mod
        self.uni_num_drug, self.uni_num_prot = uni_num_drug, uni_num_prot
        self.pp_encoder = PPEncoder(in_dim_prot)
        self.embed = Param(torch.empty(in_dim_drug, n_embed))
        self.hgcn = MyHierarchyConv(self.pp_encoder.out_dim, prot_drug_dim, uni_num_prot, uni_num_drug)
        self.hdrug = torch.zeros((uni_num_drug, self.pp_encoder.out_dim), device=device)
        d_in = n_embed + self.hgcn.out_dim if mod == 'cat' else n_embed
        self.rgcn1 = MyRGCNConv2(d_in, n_hid1, num_dd_et, num_base, after_relu=False)
        self.rgcn2 = MyRGCNConv2(n_hid1, n_hid2, num_dd_et, num_base, after_relu=True)
        self.rgcn1.paired = self.rgcn2.paired = True
        self.rgcn1.plan_share = self.rgcn2.plan_share = {}
        self._drug_feat = _FeatureInput()
        self.prune_pp_rows = True               # conv2 of the P-P encoder only for the rows the P -> D stage reads (forward())
        self.reset_parameters()

    def reset_parameters(self):
        self.embed.data.normal_()

    def fused_plans(self, x_drug, dd_edge_index, dd_range_list, d_norm, x_prot, pp_edge_index, dp_edge_index):
        """(xd, encoder.EncoderPlans) if this call can run as ONE autograd node with its own launch schedule
        (tip_amd/encoder.py: identity protein features, pruned conv2, pair-form D-D graph on both layers, no relation
        sharding), else None.  The graphs are the layers' own cached plans: both routes build and share the same ones."""
        if switches.on('TIPK_NO_ENCODER_STEP') or not self.prune_pp_rows or not _is_identity_features(x_prot):
            return None
        r1, r2 = self.rgcn1, self.rgcn2
        if r1.shard is not None or r2.shard is not None or r1.bias is not None or r2.bias is not None:
            return None
        if not (torch.is_tensor(d_norm) and d_norm.is_cuda and self.embed.is_cuda):
            return None
        rows = self.hgcn.source_rows(dp_edge_index)
        n_prot = x_prot.shape[0]
        if rows is None or rows.numel() >= n_prot or rows.numel() == 0:
            return None
        c1, c2 = self.pp_encoder.conv1, self.pp_encoder.conv2
        pd = self.hgcn.mean_sources(None, dp_edge_index, rows=rows, graph_only=True)
        if pd is None:
            return None
        xd = self._drug_feat.apply_table(x_drug, self.embed)
        n = xd.shape[0]
        pp = c1._cache.get((pp_edge_index,), lambda: gcn_norm_graph(pp_edge_index, n_prot, c1.chunk, c1.out_channels))
        pp_rows = c2._cache_rows.get((pp_edge_index, rows), lambda: gcn_norm_graph(pp_edge_index, n_prot, c2.chunk, c2.in_channels, rows))
        plans = encoder.EncoderPlans(pp, pp_rows, pd, r1.graph_for(n, dd_edge_index, dd_range_list),
                                     r2.graph_for(n, dd_edge_index, dd_range_list), self.mod == 'cat')
        if not encoder.usable(plans, xd, self.hgcn.weight, d_norm, c2.lin.weight, c1.bias, c2.bias, r1.basis, r1.att, r2.basis, r2.att):
            return None
        return xd, plans

    def forward(self, x_drug, dd_edge_index, dd_edge_type, dd_range_list, d_norm,
                x_prot, pp_edge_index, dp_edge_index, dp_range_list):
        fused = self.fused_plans(x_drug, dd_edge_index, dd_range_list, d_norm, x_prot, pp_edge_index, dp_edge_index)
        self.last_route = 'encoder_step' if fused is not None else 'per_layer'      # (bench.py / tests read it)
        if fused is not None:
            # the whole pass as one autograd node: 8 + 8 launches, scheduled by tip_amd/encoder.py
            c1, c2, r1, r2 = self.pp_encoder.conv1, self.pp_encoder.conv2, self.rgcn1, self.rgcn2
            return encoder.encoder_step(fused[0], c1.lin.weight, c1.bias, c2.lin.weight, c2.bias, self.hgcn.weight, d_norm,
                                        r1.basis, r1.att, r1.root, r2.basis, r2.att, r2.root, fused[1])
        x0 = self.mixed_drug_features(x_drug, d_norm, x_prot, pp_edge_index, dp_edge_index, dp_range_list)
        # ReLU (:547) is applied by rgcn1's last kernel and its backward mask by rgcn2's (x1 has no other consumer)
        # ... and its final slab sum runs in rgcn2's first launch, together with rgcn2's XB / X root products
        # rgcn2's pair cells ride in rgcn1's cell launch
        tok = object()                                         # "rgcn1 gathered rgcn2's cells in THIS pass"
        x1 = self.rgcn1(x0, dd_edge_index, dd_edge_type, dd_range_list, fuse_relu='gated_downstream', defer_output=True,
                        next_layer=self.rgcn2, cells_token=tok)
        return self.rgcn2(x1, dd_edge_index, dd_edge_type, dd_range_list, gate_input=True, cells_token=tok)

    def mixed_drug_features(self, x_drug, d_norm, x_prot, pp_edge_index, dp_edge_index, dp_range_list):
        """The input of the D-D layers (src/layers.py:522-539): P-P GCN x2 -> P -> D mean -> dense map, drug embedding /
        d_norm, cat | add.  (The part of the step a relation-sharded run repeats on every rank.)"""
        xd = self._drug_feat.apply_table(x_drug, self.embed)                          # x_drug @ embed
        # Only the proteins some P -> D edge starts at are ever read from the P-P encoder's output (BioSNAP: 3 640 of
        # 19 081; 294 k of its 1.29 M edges end there): conv2 aggregates -- and back-propagates through -- those rows only.
        # Exact: the rows left out feed nothing.  `PPEncoder.forward` as a module of its own still returns every row.
        rows = self.hgcn.source_rows(dp_edge_index) if self.prune_pp_rows else None
        if rows is not None and rows.numel() < x_prot.shape[0]:
            c1, c2 = self.pp_encoder.conv1, self.pp_encoder.conv2
            g_in = group_slots_for(c2.in_channels)
            if (_is_identity_features(x_prot) or not x_prot.is_sparse) and \
                    ops.gather_sum_lin_supported(c2.in_channels, c2.out_channels, g_in) and \
                    ops.lib().tipk_gather_sum_riders_supported(c2.in_channels, g_in):
                # conv2 runs aggregate-first and its transposed aggregation has 1024-thread workgroups: conv1's ReLU backward
                # and the partial rows of conv1's bias gradient are produced in THAT launch's epilogue (h1 has no other consumer)
                link = ops.GateLink() if c1.bias is not None else None
                h1 = c1(x_prot, pp_edge_index, fuse_relu='gated_downstream', link=link)
                # conv2 + the P -> D stage as ONE node with the fused launches of tip_amd/encoder.py where they apply (round 6:
                # what a relation-sharded run repeats on every rank takes 3 + 2 launches instead of 3 + 4)
                pd_g = self.hgcn.mean_sources(None, dp_edge_index, rows=rows, graph_only=True)
                n_p = x_prot.shape[0]
                g_rows = c2._cache_rows.get((pp_edge_index, rows), lambda: gcn_norm_graph(pp_edge_index, n_p, c2.chunk, c2.in_channels, rows))
                if not switches.on('TIPK_NO_ENCODER_STEP') and torch.is_tensor(d_norm) and \
                        encoder.pd_stage_usable(g_rows, pd_g, xd, self.hgcn.weight, d_norm, c2.lin.weight, c2.bias):
                    return encoder.pd_stage(h1, c2.lin.weight, c2.bias, xd, self.hgcn.weight, d_norm, g_rows, pd_g, self.mod == 'cat', link)
                h_prot = c2(h1, pp_edge_index, rows=rows, gate_input=True, link=link)     # [len(rows), hid2]
            else:
                h1 = c1(x_prot, pp_edge_index, fuse_relu=True)
                h_prot = c2(h1, pp_edge_index, rows=rows)                                  # [len(rows), hid2]
        else:
            rows = None
            h_prot = self.pp_encoder(x_prot, pp_edge_index)                           # P-P GCN x2
        pd_graph = self.hgcn.mean_sources(None, dp_edge_index, rows=rows, graph_only=True)
        if pd_graph is not None and ops.drug_mix_gather_supported(h_prot, self.hgcn.weight, d_norm):
            # P -> D mean, dense map, /d_norm and cat | add: ONE launch forward (tipk_drug_mix_gather_fwd), two backward
            x0 = ops.drug_mix_gather(xd, h_prot, self.hgcn.weight, d_norm, self.mod == 'cat', pd_graph)
            mean = None
        else:
            mean = self.hgcn.mean_sources(h_prot, dp_edge_index, rows=rows)           # P -> D mean, no cat (:526-528)
            x0 = None
        if x0 is not None:
            pass
        elif mean is not None:                                                        # dense map + /d_norm + cat|add fused
            x0 = ops.drug_mix_mm(xd, mean, self.hgcn.weight, d_norm, self.mod == 'cat')
        else:                                                                         # an edge starts at a drug row
            if self.hdrug.device != h_prot.device:
                self.hdrug = self.hdrug.to(h_prot.device)
            pd = self.hgcn(torch.cat((h_prot, self.hdrug)), dp_edge_index, dp_range_list)
            x0 = ops.drug_mix(xd, pd, d_norm, self.mod == 'cat')                      # /d_norm, cat|add
        return x0


class FMEncoderCat(FMEncoder):
    """cat-only constructor signature of `src/layers.py:403-405`."""

    def __init__(self, device, in_dim_drug, num_dd_et, in_dim_prot, uni_num_prot, uni_num_drug,
                 prot_drug_dim=16, num_base=32, n_embed=48, n_hid1=32, n_hid2=16):
        super().__init__(device, in_dim_drug, num_dd_et, in_dim_prot, uni_num_prot, uni_num_drug,
                         prot_drug_dim, num_base, n_embed, n_hid1, n_hid2, mod='cat')


# ---------------------------------------------------------------------------------------------
# A6  DistMult decoder   (src/layers.py:581-595)
# ---------------------------------------------------------------------------------------------
class MultiInnerProductDecoder(nn.Module):
    def __init__(self, in_dim, num_et):
        super().__init__()
        self.num_et, self.in_dim = num_et, in_dim
        self.weight = Param(torch.empty(num_et, in_dim))
        self.reset_parameters()

    def reset_parameters(self):
        self.weight.data.normal_(std=1 / np.sqrt(self.in_dim))

    def forward(self, z, edge_index, edge_type, sigmoid=True):
        return ops.distmult(z, self.weight, edge_index, edge_type, sigmoid)

    def objective(self, z, pos_index, neg_index, edge_type):
        """-mean log(sigma(pos)+eps) - mean log(1-sigma(neg)+eps), fused (K9+K10)."""
        return ops.distmult_objective(z, self.weight, pos_index, neg_index, edge_type)


class NNDecoder(nn.Module):
    """The paper's DR-NN decoder (reference `src/layers.py:598-637`):
    score = sigma( relu(z[u] w1_l1) . w1_l2[r] + relu(z[v] w2_l1) . w2_l2[r] ).

    The reference gathers z[u], z[v] (E x in), multiplies E x in by in x l1 twice and gathers two
    E x l1 relation rows.  Here the node-level part is done once per node (two GEMMs with fused ReLU),
    the relation-specific dot products of ALL (node, relation) pairs are two small dense GEMMs
    (N x R tables, 2.8 MB), and a triple costs two scalar reads (`tipk_pair_table_fwd`)."""

    def __init__(self, in_dim, num_uni_edge_type, l1_dim=16):
        super().__init__()
        self.l1_dim = l1_dim
        self.w1_l1 = Param(torch.empty(in_dim, l1_dim))
        self.w1_l2 = Param(torch.empty(num_uni_edge_type, l1_dim))
        self.w2_l1 = Param(torch.empty(in_dim, l1_dim))
        self.w2_l2 = Param(torch.empty(num_uni_edge_type, l1_dim))
        self.reset_parameters()

    def reset_parameters(self):
        self.w1_l1.data.normal_()
        self.w2_l1.data.normal_()
        self.w1_l2.data.normal_(std=1 / np.sqrt(self.l1_dim))
        self.w2_l2.data.normal_(std=1 / np.sqrt(self.l1_dim))

    def forward(self, z, edge_index, edge_type):
        p = torch.relu(ops.matmul(z, self.w1_l1))
        q = torch.relu(ops.matmul(z, self.w2_l1))
        s1 = ops.matmul(p, self.w1_l2.t())                       # [N, R]: every (node, relation) dot product
        s2 = ops.matmul(q, self.w2_l2.t())
        return ops.pair_table_score(s1, s2, edge_index, edge_type, sigmoid=True)

    def objective(self, z, pos_index, neg_index, edge_type):
        """-mean log(sigma(pos)+eps) - mean log(1-sigma(neg)+eps) (src/layers.py:335-340 with this decoder as
        model/ddm-nn.py:65-102 trains it), fused: the tables are formed TRANSPOSED ([R, N]: a relation's scores are one
        contiguous row), one launch evaluates every positive and negative triple and leaves both table gradients
        (`tipk_pair_table_loss`); the four weight gradients and d z follow through the four small products."""
        p = torch.relu(ops.matmul(z, self.w1_l1))
        q = torch.relu(ops.matmul(z, self.w2_l1))
        if not ops.pair_table_loss_supported(z.shape[0]):
            # node sets whose two table rows (24 B per node) exceed the kernel's LDS: scores by the table kernels, the loss as the
            # reference spells it -- still every FLOP on the device, through tipk_pair_table_fwd / _bwd
            neg = ops.unpack_pairs(neg_index) if getattr(neg_index, '_tipk_packed_pairs', False) else neg_index
            pos_s = self(z, pos_index, edge_type)
            neg_s = self(z, neg, edge_type)
            return (-torch.log(pos_s + EPS).mean() - torch.log(1 - neg_s + EPS).mean()).view(1)
        s1t = ops.matmul(self.w1_l2, p.t())                      # [R, N]
        s2t = ops.matmul(self.w2_l2, q.t())
        return ops.pair_table_objective(s1t, s2t, pos_index, neg_index, edge_type)


# ---------------------------------------------------------------------------------------------
# A9  training framework   (src/layers.py:260-375)
# ---------------------------------------------------------------------------------------------
class Setting(object):
    def __init__(self, sp_rate=0.9, lr=0.01, prot_drug_dim=16, n_embed=48, n_hid1=32, n_hid2=16, num_base=32):
        self.sp_rate, self.lr = sp_rate, lr
        self.prot_drug_dim, self.n_embed = prot_drug_dim, n_embed
        self.n_hid1, self.n_hid2, self.num_base = n_hid1, n_hid2, num_base


class TIP(nn.Module):
    """`TIP(settings, device, mod='cat', data_path='./data/data_dict.pkl')` as in the reference.

    data_path: a pickle written by the reference's `prepare.py` loads as is; if the DEFAULT file
    does not exist (or data_path is None), the BioSNAP graph bundled with this package is split with
    seed 1111; any other missing path raises FileNotFoundError.  `.data_source` says which was used.
    `data` (extension): an already built data dict.  `fused_loss=False` computes the loss with
    torch ops on the decoder scores exactly as `src/layers.py:335-340` spells it."""

    def __init__(self, settings, device, mod='cat', data_path='./data/data_dict.pkl', data=None, fused_loss=True,
                 shard=None, decoder='distmult'):
        """shard (extension): a `tip_amd.dist.RelationShard` -- this process holds only its relations'
        edges, `rgcn*.att` rows and `decoder.weight` rows; see tip_amd/dist.py.
        decoder (extension): 'distmult' (tip.py, src/layers.py:581-595) | 'nn' -- the NNDecoder of src/layers.py:598-637 in
        the same training framework, as the reference's model/ddm-nn.py:65-102 trains it (l1_dim = 16)."""
        super().__init__()
        assert mod in {'cat', 'add'} and decoder in {'distmult', 'nn'}
        if decoder == 'nn' and shard is not None:
            raise NotImplementedError('the relation-sharded step holds decoder.weight rows only (tip_amd/dist.py)')
        self.mod, self.device, self.settings, self.fused_loss = mod, device, settings, fused_loss
        self.shard, self.decoder_kind = shard, decoder
        self.data = self.__prepare_data(data_path, settings.sp_rate, data).to(device)
        self.__prepare_model()

    DEFAULT_DATA_PATH = './data/data_dict.pkl'

    def __prepare_data(self, data_path, sp_rate, data_dict):
        if data_dict is None:
            if data_path is not None and os.path.exists(data_path):
                with open(data_path, 'rb') as f:
                    data_dict = pickle.load(f)
                self.data_source = os.path.abspath(data_path)
            elif data_path is None or data_path == self.DEFAULT_DATA_PATH:
                # only the reference's DEFAULT location may be absent (this package bundles the graph
                # that file is made from); an explicitly given path that does not exist is an error,
                # as in the reference (src/layers.py:284-285 raises FileNotFoundError)
                data_dict = build_data_dict(sp_rate=0.9)
                self.data_source = 'bundled BioSNAP blob (tip_amd/data/biosnap_v1.npz), split seed 1111'
            else:
                raise FileNotFoundError(data_path)
        else:
            self.data_source = 'data dict passed by the caller'
        data_dict = dict(data_dict)
        if sp_rate != 0.9:                                                   # :290-291
            (data_dict['dd_train_idx'], data_dict['dd_train_et'], data_dict['dd_train_range'],
             data_dict['dd_test_idx'], data_dict['dd_test_et'], data_dict['dd_test_range']) = \
                process_edges(data_dict['dd_edge_index'], p=sp_rate)
        if self.shard is not None:                                           # this rank's relations only
            from .dist import shard_data_dict
            data_dict = shard_data_dict(data_dict, self.shard)
        data = Data.from_dict(data_dict)
        self._test_neg_host = None
        return data

    def __getstate__(self):
        """`torch.save(model, ...)` (tip.py:36): the embeddings of the last forward are saved as data,
        not with their autograd history."""
        st = self.__dict__.copy()
        if torch.is_tensor(st.get('embeddings')):
            st['embeddings'] = st['embeddings'].detach()
        return st

    def __encode(self):
        d = self.data
        return self.encoder(d.d_feat, d.dd_train_idx, d.dd_train_et, d.dd_train_range, d.d_norm,
                            d.p_feat, d.pp_train_indices, d.dp_edge_index, d.dp_range_list)

    def __prepare_model(self):
        d, s = self.data, self.settings
        self.test_neg_index = typed_negative_sampling(d.dd_test_idx, d.n_drug, d.dd_test_range,    # :293
                                                      pos_offset=getattr(d, 'dd_test_pos_offset', None))
        self.encoder = FMEncoder(self.device, d.n_drug_feat, d.n_dd_et, d.n_prot, d.n_prot, d.n_drug,
                                 s.prot_drug_dim, s.num_base, s.n_embed, s.n_hid1, s.n_hid2,
                                 mod=self.mod).to(self.device)
        if self.shard is not None:
            from .dist import attach_shard
            attach_shard(self.encoder, self.shard)
        with torch.no_grad():                            # initial pass (:319, with self.device); it only
            self.embeddings = self.__encode()            # fills .embeddings and the plan caches
        if self.decoder_kind == 'nn':
            self.decoder = NNDecoder(s.n_hid2, d.n_dd_et, l1_dim=16).to(self.device)
        else:
            self.decoder = MultiInnerProductDecoder(s.n_hid2, d.n_dd_et).to(self.device)

    def forward(self, neg_index=None):
        """One full-batch training objective (:328-342).  `neg_index` (extension) injects fixed
        negatives for parity tests; by default they are drawn on device every call."""
        d = self.data
        self.embeddings = self.__encode()
        pos_index = d.dd_train_idx
        if neg_index is None:
            # (the fused objective reads pairs as 32-bit words: the sampler emits that form directly -- same draws)
            neg_index = typed_negative_sampling(d.dd_train_idx, d.n_drug, d.dd_train_range,
                                                pos_offset=getattr(d, 'dd_train_pos_offset', None),
                                                packed=bool(self.fused_loss and d.n_drug <= 65535 and pos_index.shape[1] > 0))
        if not getattr(neg_index, '_tipk_packed_pairs', False):
            neg_index = neg_index.type_as(pos_index)
        if self.shard is not None:
            return self.__sharded_objective(pos_index, neg_index)
        if self.fused_loss:
            return self.decoder.objective(self.embeddings, pos_index, neg_index, d.dd_train_et)
        pos_score = self.decoder(self.embeddings, pos_index, d.dd_train_et)
        neg_score = self.decoder(self.embeddings, neg_index, d.dd_train_et)
        return -torch.log(pos_score + EPS).mean() - torch.log(1 - neg_score + EPS).mean()

    def __sharded_objective(self, pos_index, neg_index):
        """The objective over ALL ranks' triples from shard-local work: both means of
        :338-340 run over the global triple count, so a rank's local objective (means over ITS E_k
        triples) enters with weight E_k / E; one scalar all-reduce going forward, one d z all-reduce
        going back (`sum_grad_over_ranks`), d decoder.weight rows stay local."""
        from .dist import all_reduce_sum, sum_grad_over_ranks
        z = sum_grad_over_ranks(self.embeddings, self.shard)
        if pos_index.shape[1] == 0:                                          # a rank without relations
            local = z.sum() * 0.0
        elif self.fused_loss:
            local = self.decoder.objective(z, pos_index, neg_index, self.data.dd_train_et)
        else:
            pos_score = self.decoder(z, pos_index, self.data.dd_train_et)
            neg_score = self.decoder(z, neg_index, self.data.dd_train_et)
            local = -torch.log(pos_score + EPS).mean() - torch.log(1 - neg_score + EPS).mean()
        return all_reduce_sum(local * self.shard.loss_weight, self.shard)

    def pred(self, dd_idx, dd_et):
        return self.decoder(self.embeddings, dd_idx, dd_et)

    def pred_topk(self, dd_pairs, k=10, max_triples=1 << 24):
        """Serving helper (extension): for every drug pair (u, v) of
        `dd_pairs` [2, P] the k most likely side effects -> (scores [P, k], relation ids [P, k]).
        All num_et relations of a pair are scored by the decoder kernel in one launch (pairs are
        processed in slices of at most `max_triples` triples)."""
        R = self.data.n_dd_et
        k = min(int(k), R)
        pairs = dd_pairs.to(self.embeddings.device).to(torch.int64)
        P = pairs.shape[1]
        et_all = torch.arange(R, device=pairs.device)
        vals, ids = [], []
        step = max(1, max_triples // R)
        with torch.no_grad():
            for p0 in range(0, P, step):
                sl = pairs[:, p0:p0 + step]
                n = sl.shape[1]
                s = self.decoder(self.embeddings, sl.repeat_interleave(R, dim=1), et_all.repeat(n)).view(n, R)
                top = torch.topk(s, k, dim=1)
                vals.append(top.values)
                ids.append(top.indices)
        if not vals:
            return (torch.zeros((0, k), device=pairs.device), torch.zeros((0, k), dtype=torch.int64, device=pairs.device))
        return torch.cat(vals), torch.cat(ids)

    def test(self, print_output=True):
        self.eval()
        d = self.data
        with torch.no_grad():
            pos_score = self.decoder(self.embeddings, d.dd_test_idx, d.dd_test_et)
            neg_score = self.decoder(self.embeddings, self.test_neg_index, d.dd_test_et)
        if self.shard is not None:
            return self.__sharded_test(pos_score, neg_score, print_output)
        return self.compute_auprc_auroc_ap_by_et(pos_score, neg_score, d.dd_test_range, print_output)

    def __sharded_test(self, pos_score, neg_score, print_output):
        """Per-relation metrics of the local relations, all-gathered into the full [3, R] record."""
        import torch.distributed as dist
        sh = self.shard
        local = auprc_auroc_ap_by_range(pos_score, neg_score, self.data.dd_test_range) if sh.rel_ids.numel() \
            else np.zeros((3, 0))
        parts = [None] * sh.world
        if dist.is_initialized():
            dist.all_gather_object(parts, (sh.rel_ids.cpu().numpy(), np.asarray(local)), group=sh.group)
        else:
            parts = [(sh.rel_ids.cpu().numpy(), np.asarray(local))]
        record = np.zeros((3, sh.n_relations))
        for ids, rec in parts:
            record[:, ids] = rec
        if print_output and sh.rank == 0:
            auprc, auroc, ap = record.sum(axis=1) / sh.n_relations
            print('On test set: auprc:{:0.4f}   auroc:{:0.4f}   ap@50:{:0.4f}    '.format(auprc, auroc, ap))
        return record

    def compute_auprc_auroc_ap_by_et(self, pos_score, neg_score, dd_range, print_out):
        record = auprc_auroc_ap_by_range(pos_score, neg_score, dd_range)      # [3, R]
        if print_out:
            auprc, auroc, ap = record.sum(axis=1) / self.data.n_dd_et
            print('On test set: auprc:{:0.4f}   auroc:{:0.4f}   ap@50:{:0.4f}    '.format(auprc, auroc, ap))
        return record
