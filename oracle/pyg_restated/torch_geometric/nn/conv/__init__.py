"""Restated subset of torch_geometric.nn.conv (PyG 2.0.1) -- TEST INFRASTRUCTURE ONLY.

Semantics restated (published PyG 2.0.1 behaviour, flow='source_to_target'):

* ``MessagePassing.propagate(edge_index, **kw)``: arguments of ``message`` that end in ``_j`` are
  ``kw[name].index_select(0, edge_index[0])``, those ending in ``_i`` use ``edge_index[1]``; the
  special names ``edge_index``, ``edge_index_i``, ``edge_index_j`` are always available; other
  names are passed through from ``kw``.  ``aggregate`` = ``torch_scatter.scatter(msg,
  edge_index[1], dim=0, dim_size=N, reduce=aggr)`` with N = size of the lifted tensor's node
  dimension; ``mean`` = sum / clamp(count, min=1).  ``update(aggr_out, **kw)`` is then called
  with the arguments it names.
* ``GCNConv``: ``gcn_norm`` (add remaining self loops with weight 1, deg = scatter_add(w, col),
  norm = deg^-1/2[row] * w * deg^-1/2[col], inf -> 0), bias-free ``lin`` with glorot weight
  ``[out, in]``, zero ``bias`` added after aggregation, ``cached=True`` keeps the normalised graph.
"""
import inspect
import math

import torch
from torch import nn


def _scatter(src, index, dim_size, reduce):
    out = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    out.index_add_(0, index, src)
    if reduce == 'add':
        return out
    if reduce == 'mean':
        count = torch.zeros(dim_size, dtype=src.dtype, device=src.device)
        count.index_add_(0, index, torch.ones(index.numel(), dtype=src.dtype, device=src.device))
        count.clamp_(min=1)
        return out / count.view((-1,) + (1,) * (src.dim() - 1))
    raise NotImplementedError(reduce)


class MessagePassing(nn.Module):
    special_args = {'edge_index', 'edge_index_i', 'edge_index_j', 'size', 'size_i', 'size_j'}

    def __init__(self, aggr='add', flow='source_to_target', node_dim=-2):
        super().__init__()
        assert aggr in ('add', 'mean')
        assert flow == 'source_to_target'
        self.aggr = aggr
        self.flow = flow
        self.node_dim = node_dim
        self._msg_args = [p for p in inspect.signature(self.message).parameters]
        self._upd_args = [p for p in inspect.signature(self.update).parameters][1:]

    def propagate(self, edge_index, size=None, **kwargs):
        j, i = 0, 1
        dim_size = None
        coll = {}
        for arg in set(self._msg_args) | set(self._upd_args):
            if arg[-2:] in ('_i', '_j') and arg not in self.special_args:
                data = kwargs[arg[:-2]]
                if dim_size is None:
                    dim_size = data.size(0)
                coll[arg] = data.index_select(0, edge_index[j if arg.endswith('_j') else i])
            elif arg not in self.special_args:
                coll[arg] = kwargs.get(arg)
        coll['edge_index'] = edge_index
        coll['edge_index_i'] = edge_index[i]
        coll['edge_index_j'] = edge_index[j]
        msg = self.message(**{a: coll[a] for a in self._msg_args})
        out = _scatter(msg, edge_index[i], dim_size, self.aggr)
        return self.update(out, **{a: coll[a] for a in self._upd_args})

    def message(self, x_j):
        return x_j

    def update(self, aggr_out):
        return aggr_out


def gcn_norm(edge_index, num_nodes, dtype):
    row, col = edge_index[0], edge_index[1]
    w = torch.ones(row.numel(), dtype=dtype, device=row.device)
    # add_remaining_self_loops: keep non-loop edges, then one loop per node (existing loop
    # weights would be kept; the unweighted case has weight 1 either way).
    keep = row != col
    loop = torch.arange(num_nodes, dtype=row.dtype, device=row.device)
    row = torch.cat([row[keep], loop])
    col = torch.cat([col[keep], loop])
    w = torch.cat([w[keep], torch.ones(num_nodes, dtype=dtype, device=w.device)])
    deg = torch.zeros(num_nodes, dtype=dtype, device=w.device).index_add_(0, col, w)
    dis = deg.pow(-0.5)
    dis.masked_fill_(dis == float('inf'), 0)
    return torch.stack([row, col]), dis[row] * w * dis[col]


class _Linear(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels))
        a = math.sqrt(6.0 / (in_channels + out_channels))      # glorot
        self.weight.data.uniform_(-a, a)

    def forward(self, x):
        return x @ self.weight.t()


class GCNConv(MessagePassing):
    def __init__(self, in_channels, out_channels, cached=False, bias=True):
        super().__init__(aggr='add')
        self.cached = cached
        self._cached_edge_index = None
        self.lin = _Linear(in_channels, out_channels)
        self.bias = nn.Parameter(torch.zeros(out_channels))

    def forward(self, x, edge_index):
        cache = self._cached_edge_index
        if cache is None:
            ei, ew = gcn_norm(edge_index, x.size(0), self.lin.weight.dtype)
            if self.cached:
                self._cached_edge_index = (ei, ew)
        else:
            ei, ew = cache
        x = self.lin(x)
        out = self.propagate(ei, x=x, edge_weight=ew)
        return out + self.bias

    def message(self, x_j, edge_weight):
        return edge_weight.view(-1, 1) * x_j
