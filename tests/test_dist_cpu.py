"""world_size-2 gloo tests (CPU) of the relation-sharding logic in tip_amd/dist.py.

The HIP kernels cannot run here, so each rank evaluates ITS shard with the oracle's arithmetic and
the test checks that partition + shard extraction + the packed all-reduce reproduce the unsharded
layer (forward and every gradient).  The GPU counterpart (same helpers, real kernels, 2 ranks on
one device over gloo) is tests/test_gpu_layers.py::test_sharded_encoder_two_ranks.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_golden
from oracle import tip_oracle as O
from tip_amd.dist import partition_relations, shard_edges, all_reduce_packed, RelationShard


def test_partition_is_balanced_and_complete():
    sizes = [450, 51546, 4302, 900, 12000, 7000, 7000, 300, 25000, 1000]
    for world in (1, 2, 3, 8):
        parts = partition_relations(sizes, world)
        assert sorted(r for p in parts for r in p) == list(range(len(sizes)))
        loads = [sum(sizes[r] for r in p) for p in parts]
        assert max(loads) - min(loads) <= max(sizes)              # LPT bound
        assert all(p == sorted(p) for p in parts)
    assert partition_relations(sizes, 2) == partition_relations(sizes, 2)   # deterministic
    assert partition_relations([5, 5], 4)[2:] == [[], []]                     # more ranks than relations


def test_shard_edges_local_ids():
    g = load_golden('rgcn_sym')
    ei, rel = shard_edges(g['dd_idx'], g['dd_range'], [1, 3, 6])
    rg = g['dd_range']
    want = torch.cat([g['dd_idx'][:, rg[r, 0]:rg[r, 1]] for r in (1, 3, 6)], 1)
    assert torch.equal(ei, want)
    sizes = [int(rg[r, 1] - rg[r, 0]) for r in (1, 3, 6)]
    assert rel.tolist() == [0] * sizes[0] + [1] * sizes[1] + [2] * sizes[2]
    e0, r0 = shard_edges(g['dd_idx'], g['dd_range'], [])
    assert e0.shape == (2, 0) and r0.numel() == 0


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        g = load_golden('rgcn_sym', torch.float64)
        x, basis, att, root = g['x'], g['basis'], g['att'], g['root']
        n, nb, d_out = x.shape[0], basis.shape[0], basis.shape[2]
        rg = g['dd_range']
        parts = partition_relations((rg[:, 1] - rg[:, 0]).tolist(), world)
        shard = RelationShard(parts[rank], rank, world)
        ei, rel = shard_edges(g['dd_idx'], rg, shard.rel_ids)
        ids = shard.rel_ids
        # forward: partial aggregate of this rank's relations, summed over ranks, then scale + root
        xb = torch.einsum('ni,bio->bno', x, basis)
        y = (att[ids] @ xb.reshape(nb, -1)).reshape(len(ids) * n, d_out)
        agg = O.gather_sum(y, rel * n + ei[0], ei[1], n)
        all_reduce_packed([agg])
        deg = O.in_degree(g['dd_idx'][1], n, x.dtype)            # GLOBAL degree
        out = agg / deg.unsqueeze(1) + x @ root
        # backward: partial dX / d basis / d att rows, one packed collective
        up = g['upstream']
        gs = up / deg.unsqueeze(1)
        g_y = O.gather_sum(gs, ei[1], rel * n + ei[0], len(ids) * n).reshape(len(ids), n * d_out)
        g_att = torch.zeros_like(att)
        g_att[ids] = g_y @ xb.reshape(nb, -1).t()
        g_xb = (att[ids].t() @ g_y).reshape(nb, n, d_out)
        g_basis = torch.einsum('ni,bno->bio', x, g_xb)
        g_x = torch.einsum('bno,bio->ni', g_xb, basis)
        all_reduce_packed([g_x, g_basis, g_att])
        g_x = g_x + up @ root.t()
        ok = True
        for got, key in ((out, 'out'), (g_x, 'grad_x'), (g_basis, 'grad.basis'), (g_att, 'grad.att')):
            ok = ok and torch.allclose(got, g[key], rtol=1e-4, atol=1e-5)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_sharded_rgcn_equals_unsharded_world2():
    world = 2
    ctx = mp.get_context('spawn')
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(100)
        assert p.exitcode == 0
    assert dict(ret) == {0: True, 1: True}
