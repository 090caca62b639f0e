"""world_size-2 gloo tests (CPU) of the relation-sharding logic in tip_amd/dist.py.

The HIP kernels cannot run here, so each rank evaluates ITS shard with the oracle's arithmetic and
the tests check that partition + `shard_data_dict` + the shard-local parameter rows + the collectives
(flat partial aggregate, flat [dX | d basis], d z, one loss scalar) reproduce the unsharded layer and
the unsharded training objective -- forward and every gradient.  The GPU counterpart (same helpers,
real kernels, 2 ranks on one device over gloo) is tests/test_gpu_layers.py::test_sharded_training_step_two_ranks.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_golden
from oracle import tip_oracle as O
from tip_amd.dist import (partition_relations, shard_edges, make_shard, shard_data_dict, shard_state_dict,
                          all_reduce_sum, sum_grad_over_ranks, RelationShard)


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


def test_shard_data_dict_and_state_dict_rows():
    """A rank's data dict holds only its relations (local ids, local ranges, train AND test), knows the
    global in-degree / triple count, and its parameter rows are the matching rows of the full tensors."""
    from tip_amd.data import build_data_dict
    dd = build_data_dict(max_relations=7)
    R = dd['n_dd_et']
    seen = []
    total = 0
    for rank in range(3):
        sh = make_shard(dd['dd_train_range'], rank, 3)
        loc = shard_data_dict(dd, sh)
        ids = sh.rel_ids.tolist()
        seen += ids
        assert loc['n_dd_et'] == len(ids) and loc['dd_train_range'].shape == (len(ids), 2)
        assert sh.n_train_total == dd['dd_train_idx'].shape[1] and sh.n_relations == R
        assert torch.equal(sh.in_degree, torch.bincount(dd['dd_train_idx'][1], minlength=dd['n_drug']))
        total += sh.n_train_local
        for j, r in enumerate(ids):
            a, b = dd['dd_train_range'][r].tolist()
            la, lb = loc['dd_train_range'][j].tolist()
            assert torch.equal(loc['dd_train_idx'][:, la:lb], dd['dd_train_idx'][:, a:b])
            assert bool((loc['dd_train_et'][la:lb] == j).all())
            a, b = dd['dd_test_range'][r].tolist()
            la, lb = loc['dd_test_range'][j].tolist()
            assert torch.equal(loc['dd_test_idx'][:, la:lb], dd['dd_test_idx'][:, a:b])
        assert loc['pp_train_indices'] is dd['pp_train_indices']             # shared, not copied
        full = {'encoder.rgcn1.att': torch.arange(R * 2.).view(R, 2), 'decoder.weight': torch.arange(R * 3.).view(R, 3),
                'encoder.embed': torch.ones(4, 4)}
        part = shard_state_dict(full, sh)
        assert torch.equal(part['encoder.rgcn1.att'], full['encoder.rgcn1.att'][ids])
        assert torch.equal(part['decoder.weight'], full['decoder.weight'][ids]) and part['encoder.embed'].shape == (4, 4)
    assert sorted(seen) == list(range(R)) and total == dd['dd_train_idx'].shape[1]


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
        sh = make_shard(rg, rank, world)
        ids = sh.rel_ids
        ei, rel = shard_edges(g['dd_idx'], rg, ids)
        att_l = att[ids]                                         # shard-local rows
        # forward: partial aggregate of this rank's relations, summed over ranks, then scale + root
        xb = torch.einsum('ni,bio->bno', x, basis)
        y = (att_l @ xb.reshape(nb, -1)).reshape(len(ids) * n, d_out)
        agg = O.gather_sum(y, rel * n + ei[0], ei[1], n)
        sh.all_reduce(agg)
        deg = O.in_degree(g['dd_idx'][1], n, x.dtype)            # GLOBAL degree
        out = agg / deg.unsqueeze(1) + x @ root
        # backward: [partial dX | partial d basis] in ONE flat buffer, d att rows stay local
        up = g['upstream']
        gs = up / deg.unsqueeze(1)
        g_y = O.gather_sum(gs, ei[1], rel * n + ei[0], len(ids) * n).reshape(len(ids), n * d_out)
        g_att_l = g_y @ xb.reshape(nb, -1).t()
        g_xb = (att_l.t() @ g_y).reshape(nb, n, d_out)
        flat = torch.empty(x.numel() + basis.numel(), dtype=x.dtype)
        g_x, g_basis = flat[:x.numel()].view_as(x), flat[x.numel():].view_as(basis)
        g_basis.copy_(torch.einsum('ni,bno->bio', x, g_xb))
        g_x.copy_(torch.einsum('bno,bio->ni', g_xb, basis))
        sh.all_reduce(flat)
        g_x = g_x + up @ root.t()
        ok = True
        for got, want in ((out, g['out']), (g_x, g['grad_x']), (g_basis, g['grad.basis']), (g_att_l, g['grad.att'][ids])):
            ok = ok and torch.allclose(got, want, rtol=1e-4, atol=1e-5)

        # the training objective from shard-local triples: weight E_k / E, one scalar all-reduce forward,
        # the d z all-reduce backward (autograd through the two helper Functions of tip_amd/dist.py)
        gd = load_golden('decoder', torch.float64)
        z0, w = gd['z'], gd['weight']
        pos, et = gd['dd_idx'], gd['dd_et']
        gen = torch.Generator().manual_seed(3)
        neg = torch.randint(0, z0.shape[0], pos.shape, generator=gen)
        R2 = w.shape[0]
        sizes = torch.bincount(et, minlength=R2)
        end = torch.cumsum(sizes, 0)
        rg2 = torch.stack([end - sizes, end], 1)

        def objective(z, wt, p_, n_, e_):
            ps = torch.sigmoid((z[p_[0]] * z[p_[1]] * wt[e_]).sum(1))
            ns = torch.sigmoid((z[n_[0]] * z[n_[1]] * wt[e_]).sum(1))
            return -torch.log(ps + 1e-13).mean() - torch.log(1 - ns + 1e-13).mean()
        zf = z0.clone().requires_grad_(True)
        wf = w.clone().requires_grad_(True)
        full = objective(zf, wf, pos, neg, et)
        full.backward()
        sh2 = make_shard(rg2, rank, world)
        sh2.n_train_total = pos.shape[1]
        p_l, e_l = shard_edges(pos, rg2, sh2.rel_ids)
        n_l, _ = shard_edges(neg, rg2, sh2.rel_ids)
        sh2.n_train_local = p_l.shape[1]
        zs = z0.clone().requires_grad_(True)
        wl = w[sh2.rel_ids].clone().requires_grad_(True)
        local = objective(sum_grad_over_ranks(zs, sh2), wl, p_l, n_l, e_l)
        loss = all_reduce_sum(local * sh2.loss_weight, sh2)
        loss.backward()
        ok = ok and torch.allclose(loss, full.detach(), rtol=1e-10, atol=1e-12)
        ok = ok and torch.allclose(zs.grad, zf.grad, rtol=1e-8, atol=1e-12)
        ok = ok and torch.allclose(wl.grad, wf.grad[sh2.rel_ids], rtol=1e-8, atol=1e-12)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_sharded_rgcn_and_objective_equal_unsharded_world2():
    world = 2
    ctx = mp.get_context('spawn')
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
    from conftest import start_ranks
    start_ranks(procs)
    for p in procs:
        p.join(100)
        assert p.exitcode == 0
    assert dict(ret) == {0: True, 1: True}
