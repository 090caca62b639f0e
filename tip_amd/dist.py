"""Relation-sharded multi-GPU execution of the D-D stage (one process per GPU, RCCL over xGMI).

The reference is single-GPU (`README.md:58`).  The D-D R-GCN layer is a sum over relations of
independent partial aggregates of the same N x d output, so it shards by relation id
(BASELINE.json north_star):

  * every rank holds all parameters (replicas) and the full edge tensors it is handed through
    the reference `forward()` signature, but builds gather plans only for ITS relations
    (greedy edge-count balancing: relation sizes span 450 ... 51 546 edges);
  * forward: partial `sum_{r in shard} A_r Y_r`  --all-reduce(sum)-->  x 1/deg(global) + X root;
  * backward: partial dX, d basis and the shard's rows of d att are packed into ONE buffer and
    all-reduced (one collective per layer and direction; messages are 40 KB - 0.5 MB, i.e.
    latency-bound on BioSNAP: the synthetic 50 M-edge graph is the scaling case);
  * the P-P and P->D stages (< 2 % of the work) are computed redundantly on every rank, so their
    gradients are identical everywhere and need no collective.

`torch.distributed` backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests of this logic.
"""
import torch
import torch.distributed as dist


def partition_relations(sizes, world):
    """Greedy longest-processing-time assignment of relations to `world` ranks by edge count.
    Returns a list (per rank) of ascending relation-id lists; deterministic on every rank."""
    sizes = [int(s) for s in sizes]
    order = sorted(range(len(sizes)), key=lambda r: (-sizes[r], r))
    load = [0] * world
    parts = [[] for _ in range(world)]
    for r in order:
        k = min(range(world), key=lambda i: (load[i], i))
        parts[k].append(r)
        load[k] += sizes[r]
    return [sorted(p) for p in parts]


class RelationShard(object):
    """This rank's share of the relations plus the process group used for the partial sums."""

    def __init__(self, rel_ids, rank, world, group=None):
        self.rel_ids = torch.as_tensor(rel_ids, dtype=torch.int64)
        self.rank, self.world, self.group = rank, world, group

    def rel_ids_on(self, device):
        if self.rel_ids.device != device:
            self.rel_ids = self.rel_ids.to(device)
        return self.rel_ids


def shard_edges(edge_index, range_list, rel_ids):
    """Edges of the relations in `rel_ids` (ascending), concatenated, with LOCAL relation ids
    0..len(rel_ids)-1 per edge.  -> (edge_index_local [2, E_k], rel_local [E_k])."""
    rg = torch.as_tensor(range_list).to(torch.int64).cpu()
    ids = [int(r) for r in torch.as_tensor(rel_ids).tolist()]
    if not ids:
        return edge_index[:, :0], torch.zeros(0, dtype=torch.int64, device=edge_index.device)
    blocks = [edge_index[:, int(rg[r, 0]):int(rg[r, 1])] for r in ids]
    sizes = torch.tensor([b.shape[1] for b in blocks], dtype=torch.int64)
    rel_local = torch.repeat_interleave(torch.arange(len(ids)), sizes).to(edge_index.device)
    return torch.cat(blocks, dim=1), rel_local


def all_reduce_packed(tensors, group=None):
    """Sum-all-reduce several tensors with ONE collective (flat pack / unpack, in place)."""
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t))
        off += n
    return tensors


def shard_encoder(encoder, range_list, rank, world, group=None):
    """Attach the same relation shard to both R-GCN layers of an `FMEncoder`."""
    rg = torch.as_tensor(range_list).to(torch.int64).cpu()
    parts = partition_relations((rg[:, 1] - rg[:, 0]).tolist(), world)
    shard = RelationShard(parts[rank], rank, world, group)
    for layer in (encoder.rgcn1, encoder.rgcn2):
        layer.shard = shard
        layer._cache.key = None                 # plans of an earlier (unsharded) call are stale
    return shard
