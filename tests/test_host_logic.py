"""CPU tests of the host side: plans, edge bookkeeping, metrics, data ingest, C-ABI surface."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden
from oracle import tip_oracle as O
from tip_amd import _lib, utils
from tip_amd.data import build_data_dict, synthetic_data_dict, Data
from tip_amd.plan import build_gather_plan, execute_plan_reference


# ------------------------------------------------------------------ plans
@pytest.mark.parametrize('chunk', [1, 4, 128])
@pytest.mark.parametrize('weighted', [False, True])
def test_gather_plan_semantics(chunk, weighted):
    g = torch.Generator().manual_seed(chunk)
    n_out, n_tab, E, d = 23, 31, 500, 6
    out_row = torch.randint(0, n_out - 3, (E,), generator=g)          # last rows empty
    out_row[:200] = 5                                                 # one very heavy row
    tab_row = torch.randint(0, n_tab, (E,), generator=g)
    w = torch.rand(E, generator=g) if weighted else None
    table = torch.randn(n_tab, d, generator=g, dtype=torch.float64)
    plan = build_gather_plan(out_row, tab_row, n_out, n_tab, w, chunk)
    got = execute_plan_reference(plan, table)
    want = O.gather_sum(table, tab_row, out_row, n_out, None if w is None else w.double())
    torch.testing.assert_close(got, want, rtol=1e-6, atol=1e-9)
    it = plan.items.long()
    lens = it[:, 1] - it[:, 0]
    assert int(lens.max()) <= chunk and bool((lens[:-1] >= lens[1:]).all())      # sorted by length
    assert int(lens.sum()) == E
    # every output row is covered: direct rows + split rows partition range(n_out)
    direct_rows = set(it[it[:, 3] == 1, 2].tolist())
    split_rows = set(plan.split_rows[:, 0].tolist())
    assert direct_rows | split_rows == set(range(n_out)) and not (direct_rows & split_rows)
    # slots of split rows tile [0, n_slots)
    if plan.n_slots:
        sr = plan.split_rows.long()
        assert int((sr[:, 2] - sr[:, 1]).sum()) == plan.n_slots
    assert plan.items.dtype == torch.int32 and plan.row_id.dtype == torch.int32


@pytest.mark.parametrize('chunk,G', [(16, 32), (4, 8), (1, 128), (64, 4)])
def test_grouped_gather_plan_semantics(chunk, G):
    """group_slots: pieces of a split row are consecutive inside one aligned block, leader first."""
    from tip_amd.plan import ITEM_DIRECT, ITEM_LEADER, ITEM_NULL, ITEM_PIECE, group_slots_for, pack_blocks
    g = torch.Generator().manual_seed(chunk * 100 + G)
    n_out, n_tab, E, d = 57, 31, 3000, 6
    out_row = torch.randint(0, n_out - 3, (E,), generator=g)
    out_row[:1200] = 5                                                # hub row: more than G chunks -> longer pieces
    tab_row = torch.randint(0, n_tab, (E,), generator=g)
    w = torch.rand(E, generator=g)
    table = torch.randn(n_tab, d, generator=g, dtype=torch.float64)
    plan = build_gather_plan(out_row, tab_row, n_out, n_tab, w, chunk, group_slots=G)
    assert plan.group_slots == G and plan.n_slots == 0 and plan.split_rows.shape[0] == 0
    got = execute_plan_reference(plan, table)                         # also asserts block containment
    want = O.gather_sum(table, tab_row, out_row, n_out, w.double())
    torch.testing.assert_close(got, want, rtol=1e-6, atol=1e-9)
    it = plan.items.long()
    fl = it[:, 3]
    assert int((it[:, 1] - it[:, 0]).sum()) == E                      # every edge exactly once
    n_grouped = int(((fl & (ITEM_PIECE | ITEM_NULL)) != 0).sum())
    assert n_grouped % G == 0 and bool(((fl[:n_grouped] & ITEM_DIRECT) == 0).all())
    assert bool((fl[n_grouped:] == ITEM_DIRECT).all())
    lens = (it[:, 1] - it[:, 0])[n_grouped:]
    assert bool((lens[:-1] >= lens[1:]).all())                        # direct items sorted by length
    lead = it[(fl & ITEM_LEADER) != 0]
    rows = set(lead[:, 2].tolist()) | set(it[n_grouped:, 2].tolist())
    assert rows == set(range(n_out)) and lead.shape[0] + (it.shape[0] - n_grouped) == n_out
    assert int((lead[:, 3] >> 8).max()) <= G
    # packing helper: no block overflows, rows do not overlap
    block, offset, n_blocks = pack_blocks([5, 3, 3, 8, 2, 7, 1], 8)
    used = {}
    for b, o, p in zip(block, offset, [5, 3, 3, 8, 2, 7, 1]):
        assert o + p <= 8 and all(not (o < o2 + p2 and o2 < o + p) for o2, p2 in used.get(b, []))
        used.setdefault(b, []).append((o, p))
    assert n_blocks == 4                                              # 29 slots of 8: best fit is optimal here
    assert [group_slots_for(d) for d in (4, 16, 32, 64, 128, 256, 3, 50)] == [128, 128, 128, 64, 32, 16, 128, 16]


def test_gather_plan_empty_and_bounds():
    plan = build_gather_plan(torch.zeros(0, dtype=torch.long), torch.zeros(0, dtype=torch.long), 4, 3)
    assert plan.items.shape == (4, 4) and plan.n_slots == 0
    assert execute_plan_reference(plan, torch.ones(3, 2)).abs().sum() == 0
    with pytest.raises(IndexError):
        build_gather_plan(torch.tensor([0, 9]), torch.tensor([0, 1]), 4, 3)


# ------------------------------------------------------------------ edge bookkeeping (vs the reference's recorded split)
def test_process_edges_replays_reference_split():
    """Same legacy generator state -> same six tensors as the reference produced (golden)."""
    g = load_golden('tip_add_small')
    rng = np.random.RandomState(18)
    n_drug = g['n_drug']
    raw = []
    for s in (30, 25, 120, 60):                       # the generator recipe of oracle/make_golden.py
        u = rng.randint(0, n_drug, s)
        v = rng.randint(0, n_drug, s)
        raw.append(torch.from_numpy(np.stack([np.minimum(u, v), np.maximum(u, v)]).astype(np.int64)))
    out = utils.process_edges(raw, rng=np.random.RandomState(18))
    for got, key in zip(out, ['dd_train_idx', 'dd_train_et', 'dd_train_range', 'dd_test_idx', 'dd_test_et',
                              'dd_test_range']):
        assert torch.equal(got, g[key]), key


def test_bidirection_helpers():
    e = torch.tensor([[5, 1, 2], [3, 4, 2]])
    both = utils.to_bidirection(e)
    assert both.tolist() == [[5, 1, 2, 3, 4, 2], [3, 4, 2, 5, 1, 2]]
    kept = utils.remove_bidirection(both)
    assert kept.tolist() == [[5, 4], [3, 1]]
    sp = utils.sparse_id(4)
    assert sp.is_sparse and torch.equal(sp.to_dense(), torch.eye(4))


# ------------------------------------------------------------------ metrics vs sklearn (the reference's estimator)
@pytest.mark.parametrize('ties', [False, True])
def test_metrics_match_sklearn(ties):
    rng = np.random.RandomState(3)
    y = (rng.rand(400) < 0.4).astype(np.float64)
    s = rng.rand(400) + 0.3 * y
    if ties:
        s = np.round(s, 1)
    got = utils.auprc_auroc_ap(y, s)
    want = O.auprc_auroc_ap(y, s)
    np.testing.assert_allclose(got, want, rtol=1e-10)


def test_metrics_by_range_matches_reference_record():
    g = load_golden('tip_add_small')
    p = {k[len('encoder.'):]: v for k, v in g.items() if isinstance(k, str) and k.startswith('encoder.')}
    data = dict(dd_train_idx=g['dd_train_idx'], dd_train_range=g['dd_train_range'], d_norm=g['d_norm'],
                pp_train_indices=g['pp_train_indices'], dp_edge_index=g['dp_edge_index'],
                n_drug=g['n_drug'], n_prot=g['n_prot'])
    z, _ = O.fm_encoder_fwd(p, data, 'add')
    ps = O.distmult_fwd(z, g['dd_test_idx'], g['dd_test_et'], g['decoder.weight'])
    ns = O.distmult_fwd(z, g['test_neg'], g['dd_test_et'], g['decoder.weight'])
    rec = utils.auprc_auroc_ap_by_range(ps, ns, g['dd_test_range'])
    np.testing.assert_allclose(rec, g['record'].numpy(), rtol=1e-6)


# ------------------------------------------------------------------ data ingest
def test_biosnap_data_dict_schema_and_sizes():
    d = build_data_dict(max_relations=5)
    assert d['n_drug'] == 645 and d['n_prot'] == 19081 and d['n_dd_et'] == 5
    E = d['dd_train_idx'].shape[1]
    assert d['dd_train_et'].shape == (E,) and d['dd_train_range'].shape == (5, 2)
    assert int(d['dd_train_range'][-1, 1]) == E
    for r, (a, b) in enumerate(d['dd_train_range'].tolist()):
        blk = d['dd_train_idx'][:, a:b]
        h = (b - a) // 2
        assert torch.equal(blk[:, :h], blk[:, h:].flip(0))           # [u<v half | mirrored half]
        assert bool((blk[0, :h] < blk[1, :h]).all())
        assert bool((d['dd_train_et'][a:b] == r).all())
    assert d['pp_train_indices'].shape[0] == 2 and d['dp_edge_index'].shape == (2, 18596)
    assert int(d['dp_edge_index'][1].min()) >= d['n_prot']
    assert d['d_feat'].is_sparse and d['d_norm'].shape == (645,)
    # same seed -> same split; different seed -> different split
    assert torch.equal(build_data_dict(max_relations=5)['dd_train_idx'], d['dd_train_idx'])
    assert build_data_dict(max_relations=5, seed=1)['dd_train_idx'].shape != d['dd_train_idx'].shape or \
        not torch.equal(build_data_dict(max_relations=5, seed=1)['dd_train_idx'], d['dd_train_idx'])
    moved = Data.from_dict(d).to('cpu')
    assert moved.n_drug == 645 and isinstance(moved.dd_edge_index, list)


def test_synthetic_graph_shape():
    d = synthetic_data_dict(n_drug=200, n_rel=17, n_edges=20000, seed=3)
    E = d['dd_train_idx'].shape[1]
    assert E == 20000 and d['dd_train_range'].shape == (17, 2) and int(d['dd_train_range'][-1, 1]) == E
    assert bool((d['dd_train_idx'][0] != d['dd_train_idx'][1]).all())
    sizes = (d['dd_train_range'][:, 1] - d['dd_train_range'][:, 0])
    assert int(sizes.min()) >= 2 and int(sizes.max()) > 4 * int(sizes.median())    # skewed


# ------------------------------------------------------------------ C ABI surface
def _declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'tipk.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(tipk_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    names = _declared_symbols()
    assert names and set(names) == set(_lib.SIGNATURES), (names, sorted(_lib.SIGNATURES))
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), n
    L = _lib.lib()
    assert L.tipk_abi_version() == _lib.ABI_VERSION
    assert L.tipk_strerror(0) == b'ok' and b'invalid' in L.tipk_strerror(-1)


def test_product_never_imports_the_oracle():
    """The product path must not route through the oracle (or any CPU fallback)."""
    pkg = os.path.join(ROOT, 'tip_amd')
    for fn in os.listdir(pkg):
        if fn.endswith('.py'):
            src = open(os.path.join(pkg, fn)).read()
            assert 'import oracle' not in src and 'from oracle' not in src, fn


def test_ops_refuse_cpu_tensors():
    from tip_amd import ops
    plan = build_gather_plan(torch.tensor([0, 1]), torch.tensor([1, 0]), 2, 2)
    with pytest.raises(_lib.TipkError):
        ops.gather_sum(plan, torch.ones(2, 4))


# ------------------------------------------------------------------ relation-local plans (LDS kernels)
def test_rel_plan_semantics():
    from tip_amd.plan import build_rel_plan, execute_rel_plan_reference, assign_relations
    g = torch.Generator().manual_seed(4)
    N, R, E, d = 29, 7, 900, 4
    rel = torch.sort(torch.randint(0, R - 1, (E,), generator=g)).values          # last relation empty
    src = torch.randint(0, N, (E,), generator=g)
    dst = torch.randint(0, N - 3, (E,), generator=g)
    dst[:300] = 2                                                              # a hub node
    y = torch.randn(R * N, d, generator=g, dtype=torch.float64)
    want = torch.zeros(N, d, dtype=torch.float64).index_add_(0, dst, y[rel * N + src])
    gp = torch.randn(N, d, generator=g, dtype=torch.float64)
    wantb = torch.zeros(R * N, d, dtype=torch.float64).index_add_(0, rel * N + src, gp[dst])
    for max_unit in (10 ** 9, 64, 8):                                         # one unit per relation ... many shares
        plan = build_rel_plan(dst, src, rel, N, R, n_wg=4, max_unit=max_unit)
        torch.testing.assert_close(execute_rel_plan_reference(plan, y, False), want)
        planb = build_rel_plan(src, dst, rel, N, R, n_wg=4, backward=True, max_unit=max_unit)
        torch.testing.assert_close(execute_rel_plan_reference(planb, gp, True), wantb)   # + every row written once
        U = plan.n_units
        assert plan.idx.dtype == torch.uint16 and plan.runs.shape == (U, N, 2) and plan.node_at.shape == (U, N)
        assert sorted(plan.wg_rels.tolist()) == list(range(U)) and plan.wg_rel_ptr.tolist()[-1] == U
        assert plan.unit_rel.tolist() == sorted(plan.unit_rel.tolist()) and set(plan.unit_rel.tolist()) == set(range(R))
        assert int(plan.node_at[0, 0]) == 2                                   # hub first in relation 0's first unit
        meta = plan.unit_meta.long()                                          # descriptors in workgroup order
        assert meta.shape == (U, 8) and meta[:, 0].tolist() == plan.wg_rels.tolist()
        assert meta[:, 1].tolist() == plan.unit_rel[meta[:, 0]].tolist()
        assert meta[:, 2].tolist() == plan.unit_npos[meta[:, 0]].tolist()
        assert meta[:, 3].tolist() == plan.rel_len[meta[:, 0]].tolist()
        assert (meta[:, 4] + (meta[:, 5] << 32)).tolist() == plan.rel_idx_off[meta[:, 0]].tolist()
        if max_unit == 10 ** 9:
            assert U == R and planb.unit_npos.tolist() == [N] * R
        else:
            assert U > R and int(plan.rel_len.max()) <= max(max_unit, 304) + 8 * N   # hub run of 300 edges cannot be cut
    # launch-shape-aware plans: ids of a run reordered for conflict-free LDS reads + pre-scaled offsets
    # (same sums -- the order inside a run only fixes the order of the fp32 additions)
    from tip_amd.plan import bank_rotation
    for lanes in (1, 2, 4, 8, 16):
        for max_unit in (10 ** 9, 64):
            plan = build_rel_plan(dst, src, rel, N, R, n_wg=4, max_unit=max_unit, lanes=lanes)
            torch.testing.assert_close(execute_rel_plan_reference(plan, y, False), want)
            planb = build_rel_plan(src, dst, rel, N, R, n_wg=4, backward=True, max_unit=max_unit, lanes=lanes)
            torch.testing.assert_close(execute_rel_plan_reference(planb, gp, True), wantb)
            unit = plan.idx_unit
            assert unit == min(lanes * 16, 2048) and int(plan.idx.to(torch.int32).max()) == N * unit <= 65535
        n_cls, rot = bank_rotation(lanes)
        assert n_cls == max(1, 16 // lanes) and len(rot) == 64 // lanes
    # L = 4: inside a run the classes (node mod 4) are sorted starting at the slot's rotation
    plan = build_rel_plan(dst, src, rel, N, R, n_wg=4, lanes=4)
    ids = (plan.idx.to(torch.int64) // plan.idx_unit)
    n_cls, rot = bank_rotation(4)
    for u in range(plan.n_units):
        e0 = int(plan.rel_idx_off[u])
        for p in range(int(plan.unit_npos[u])):
            b, ln = plan.runs[u, p].tolist()
            run = ids[e0 + b:e0 + b + ln]
            run = run[run < N]
            slot = p % 256                                              # band 0 of a 256-slot workgroup (N < 256 positions)
            key = (run % n_cls - rot[slot % 16]) % n_cls
            assert bool((key[1:] >= key[:-1]).all())
    # wave-stream plan of the transposed pass: every row exactly once, runs continue in their slot, balanced streams
    from tip_amd.plan import build_stream_plan, execute_stream_plan_reference
    for lanes in (1, 2, 4, 8, 16):
        sp = build_stream_plan(src, dst, rel, N, R, 2, lanes, piece=4)
        torch.testing.assert_close(execute_stream_plan_reference(sp, gp), wantb)
        assert sp.cells.shape[1] == 64 // lanes and sp.ids.numel() == sp.n_bands * 4 * (64 // lanes) * 8
        assert int(sp.ids.to(torch.int32).max()) <= N * sp.idx_unit <= 65535
        per_wave = (sp.wave_ptr[1:] - sp.wave_ptr[:-1])
        assert int(per_wave.sum()) == sp.n_bands and int(per_wave.max()) - int(per_wave.min()) <= max(4, sp.n_bands // 8)
    # pair form of the forward pass: rows = (source, destination) cells, table = att; a symmetric graph builds the
    # cells with source <= destination from half the edges and mirrors them
    from tip_amd.plan import build_stream_plan_rows
    att = torch.randn(R, 32, generator=g, dtype=torch.float64)
    s2, d2, r2 = torch.cat([src, dst]), torch.cat([dst, src]), torch.cat([rel, rel])      # every relation symmetric
    full = torch.zeros(N * N, 32, dtype=torch.float64).index_add_(0, s2 * N + d2, att[r2])
    keep = s2 <= d2
    half = build_stream_plan_rows(s2[keep] * N + d2[keep], r2[keep], N * N, R, 2, 8, piece=4)
    assert half.n_edges == int(keep.sum()) and half.n_table == R
    c = execute_stream_plan_reference(half, att).view(N, N, 32)
    mirrored = c + torch.triu(c.permute(2, 0, 1), 1).permute(2, 1, 0)
    torch.testing.assert_close(mirrored.view(N * N, 32), full)
    sp1 = build_stream_plan(src, dst, rel, N, R, 1, 4, piece=2)                 # shorter cells: more continuation bands
    torch.testing.assert_close(execute_stream_plan_reference(sp1, gp), wantb)
    # compact node-major rows (tipk.h section 2d): only the (node, relation) pairs with edges, grouped by node
    spc = build_stream_plan(src, dst, rel, N, R, 2, 8, piece=4, compact=True)
    cr = spc.compact
    got = execute_stream_plan_reference(spc, gp)                                # asserts: every compact row written exactly once
    cnt = torch.bincount(rel * N + src, minlength=R * N).view(R, N)
    assert cr.n_rows == int((cnt > 0).sum()) == got.shape[0] and cr.pos.shape == (N, -(-R // 64) * 64)
    pos = cr.pos.long()
    wb = wantb.view(R, N, -1)
    for u in range(N):
        rows = torch.arange(int(cr.node_ptr[u]), int(cr.node_ptr[u + 1]))
        rels = cr.row_rel[rows].long()
        assert bool((rels[1:] > rels[:-1]).all()) and torch.equal(rels, torch.nonzero(cnt[:, u] > 0).flatten())
        assert torch.equal(pos[u, rels], rows)
        torch.testing.assert_close(got[rows], wb[rels, u])
    assert int((pos == cr.n_rows).sum()) == N * pos.shape[1] - cr.n_rows
    nd = cr.node_desc.long()
    assert sorted(nd[:, 0].tolist()) == list(range(N)) and bool(((nd[1:, 2] - nd[1:, 1]) <= (nd[:-1, 2] - nd[:-1, 1])).all())
    assert torch.equal(nd[:, 1], cr.node_ptr.long()[nd[:, 0]]) and torch.equal(nd[:, 2], cr.node_ptr.long()[nd[:, 0] + 1])
    ptr, rels = assign_relations([10, 1, 7, 7, 3], 2, fixed_cost=0)
    loads = [sum([10, 1, 7, 7, 3][r] for r in rels[ptr[i]:ptr[i + 1]].tolist()) for i in range(2)]
    assert sorted(loads) == [14, 14]


def test_relation_tasks_and_sampler_keys_host_side():
    """Decoder task table (tip_amd/ops.py) and the host mirror of the device sampler-key function."""
    from tip_amd import ops, neg_sampling as NS
    from oracle.philox_sampler import call_key
    et = torch.repeat_interleave(torch.arange(5), torch.tensor([10000, 0, 3, 4097, 1]))
    tasks = ops.relation_tasks(et)
    assert tasks.dtype == torch.int32 and tasks.shape[1] == 4 and bool((tasks[:, 3] == 1).all())
    t = tasks.long()[:, :3]
    assert int((t[:, 2] - t[:, 1]).sum()) == et.numel() and int((t[:, 2] - t[:, 1]).max()) <= ops.TASK_POSITIONS
    assert bool(((t[:-1, 2] - t[:-1, 1]) >= (t[1:, 2] - t[1:, 1])).all())        # largest first
    for r, a, b in t.tolist():
        assert bool((et[a:b] == r).all())
    covered = torch.zeros(et.numel(), dtype=torch.int32)
    for r, a, b in t.tolist():
        covered[a:b] += 1
    assert bool((covered == 1).all())
    assert ops.relation_tasks(torch.tensor([0, 1, 0, 1])) is None                 # not grouped by relation
    # mirrored positives ([pairs | flipped pairs] per relation, src/utils.py:35-65): first halves weigh 2, the rest 0
    half = [torch.tensor([[0, 1, 2], [3, 4, 5]]), torch.tensor([[7], [2]]), torch.zeros((2, 0), dtype=torch.long)]
    pos = torch.cat([torch.cat([h, h.flip(0)], dim=1) for h in half], dim=1)
    et2 = torch.repeat_interleave(torch.tensor([0, 2, 5]), torch.tensor([6, 2, 0]))
    tk = ops.relation_tasks(et2, pos).long()
    assert sorted(tk.tolist()) == sorted([[0, 0, 3, 2], [0, 3, 6, 0], [2, 6, 7, 2], [2, 7, 8, 0]])
    broken = pos.clone(); broken[0, 4] = 9                                        # one pair not mirrored: plain weights
    assert bool((ops.relation_tasks(et2, broken)[:, 3] == 1).all())
    for seed, n in ((0, 0), (1111, 7), ((1 << 64) - 1, 123456)):
        assert NS.call_key(seed, n) == call_key(seed, n)
    assert len({call_key(5, n) for n in range(100)}) == 100


# ------------------------------------------------------------------ hygiene
def test_library_matches_its_sources_and_reads_no_environment():
    """The loaded .so was compiled from the sources next to it (content digest, not mtime), and no
    kernel launch path reads the environment: switches are explicit options, and the compute-skipping
    debug switches are refused by the release build."""
    assert _lib.build_id() == _lib.source_digest()
    csrc = os.path.join(ROOT, 'tip_amd', 'csrc')
    for fn in os.listdir(csrc):
        if fn.endswith(('.hip', '.cpp', '.h')):
            assert 'getenv' not in open(os.path.join(csrc, fn)).read(), fn
    assert _lib.get_option('gemm_no_stream') == 0
    _lib.set_option('gemm_no_stream', 1)
    assert _lib.get_option('gemm_no_stream') == 1
    _lib.set_option('gemm_no_stream', 0)
    for name in ('rg_debug', 'dp_debug'):
        with pytest.raises(_lib.TipkError):
            _lib.set_option(name, 1)                                   # release build: no skip code inside
        _lib.set_option(name, 0)
    with pytest.raises(_lib.TipkError):
        _lib.set_option('no_such_option', 1)


def test_modules_pickle_after_caching_plans():
    """`torch.save(model, ...)` as the reference's tip.py:36: plan caches (device arrays + closures) are
    derived state and are dropped by pickling / deepcopy; they rebuild on first use."""
    import copy
    import io
    from tip_amd.layers import MyRGCNConv2, GCNConv, MyHierarchyConv, FMEncoder
    enc = FMEncoder(torch.device('cpu'), 9, 3, 11, 11, 9, prot_drug_dim=4, num_base=2, n_embed=4, n_hid1=4, n_hid2=4)
    for m in enc.modules():
        if hasattr(m, '_cache'):
            m._cache.key, m._cache.value, m._cache.pins = ('k',), (lambda: 1), (torch.zeros(1),)
    buf = io.BytesIO()
    torch.save(enc, buf)
    buf.seek(0)
    back = torch.load(buf, weights_only=False)
    for k, v in enc.state_dict().items():
        assert torch.equal(back.state_dict()[k], v)
    assert all(m._cache.value is None for m in back.modules() if hasattr(m, '_cache'))
    assert copy.deepcopy(enc).rgcn1._cache.value is None
    assert enc.rgcn1._cache.value is not None                         # the original keeps its cache


def test_tip_missing_explicit_data_path_raises():
    """Only the reference's DEFAULT pickle location may be absent (bundled graph); a mistyped explicit
    path must not silently train on other data (reference: FileNotFoundError, src/layers.py:284)."""
    from tip_amd.layers import TIP, Setting
    with pytest.raises(FileNotFoundError):
        TIP(Setting(), torch.device('cpu'), data_path='/nonexistent/data_dict.pkl')


def test_triple_validation_is_cached_and_raises():
    from tip_amd import ops
    ei = torch.tensor([[0, 1, 2], [2, 1, 0]])
    et = torch.tensor([0, 0, 1])
    ops.validate_triples(ei, et, 3, 2)
    # the verdict lives in a side table keyed by the tensor object (weakly): nothing is pinned, nothing rides in torch.save
    assert ops._facts(ei).get('range_ok') == 3 and ops._facts(et).get('range_ok') == 2
    assert not hasattr(ei, '_tipk_range_ok') and not ei.__dict__
    ei[0, 0] = 1                                                       # modified in place: checked again
    assert ops._facts(ei).get('range_ok') is None
    ops.validate_triples(ei, et, 3, 2)
    import io
    import pickle
    buf = io.BytesIO()
    torch.save(ei, buf)
    back = torch.load(io.BytesIO(buf.getvalue()))
    assert ops._facts(back).get('range_ok') is None                    # an unpickled tensor is validated again
    n_before = len(ops._TENSOR_FACTS)
    del back
    import gc
    gc.collect()
    assert len(ops._TENSOR_FACTS) == n_before - 1                      # the entry died with its tensor
    with pytest.raises(ValueError):
        ops.validate_triples(ei, et[:2], 3, 2)                         # one relation id per triple
    with pytest.raises(IndexError):
        ops.validate_triples(ei, et, 2, 2)                             # node id 2 of 2 nodes
    with pytest.raises(IndexError):
        ops.validate_triples(ei, et, 3, 1)                             # relation id 1 of 1 relation
    with pytest.raises(IndexError):
        ops.validate_triples(torch.tensor([[0, -1], [0, 0]]), None, 3, 1)


def test_reference_pickle_fixture_schema():
    """tests/golden/data_dict_small.pkl (written by the reference's prepare.py pipeline, see
    oracle/make_pickle_fixture.py) has the schema of SURVEY 8(a) A0 and `tip_amd.data.Data` takes it as is."""
    import pickle
    from tip_amd.data import Data
    with open(os.path.join(ROOT, 'tests', 'golden', 'data_dict_small.pkl'), 'rb') as f:
        dd = pickle.load(f)
    for k in ('d_feat', 'p_feat', 'dd_train_idx', 'dd_train_et', 'dd_train_range', 'dd_test_idx', 'dd_test_et',
              'dd_test_range', 'd_norm', 'pp_train_indices', 'dp_edge_index', 'dp_range_list', 'n_drug', 'n_prot',
              'n_dd_et', 'n_drug_feat', 'dd_adj_list', 'dp_adj', 'pp_adj', 'dd_edge_index'):
        assert k in dd, k
    assert dd['dd_train_idx'].dtype == torch.int64 and dd['dd_train_range'].shape == (dd['n_dd_et'], 2)
    assert dd['dp_range_list'].dtype == torch.float32 and dd['d_feat'].is_sparse
    import scipy.sparse as sp
    assert sp.issparse(dd['pp_adj']) and sp.issparse(dd['dd_adj_list'][0])          # the unused fields came along
    d = Data.from_dict(dd).to(torch.device('cpu'))
    assert d.n_drug == 645 and d.dd_train_et.numel() == d.dd_train_idx.shape[1]
    # every relation block is [pairs | mirrored pairs] (src/utils.py:53)
    for a, b in d.dd_train_range.tolist():
        h = (b - a) // 2
        assert torch.equal(d.dd_train_idx[:, a:a + h], d.dd_train_idx[:, a + h:b].flip(0))


# ------------------------------------------------------------------ device split: host-side spec
def test_split_spec_layout_and_distribution_next_to_the_reference_draw():
    """oracle/philox_split.py (the bit-exact spec of tipk_split_*): block layout = `to_bidirection` of the
    kept pairs in list order; the kept fraction is Binomial(n, p) like the reference's numpy draw
    (src/utils.py:45), held to the same z-score bar."""
    from oracle.philox_split import process_edges_spec, split_flags_spec
    from tip_amd.utils import process_edges
    rng = np.random.RandomState(0)
    sizes = [0, 5000, 1, 300, 20000]
    ptr = np.r_[0, np.cumsum(sizes)]
    pairs = np.stack([rng.randint(0, 600, ptr[-1]), rng.randint(0, 600, ptr[-1])]).astype(np.int64)
    tr, tr_et, tr_rg, te, te_et, te_rg = process_edges_spec(pairs, ptr, 0.9, 1111)
    take = split_flags_spec(ptr[-1], 0.9, 1111)
    for r in range(len(sizes)):
        a, b = ptr[r], ptr[r + 1]
        kept = pairs[:, a:b][:, take[a:b]]
        blk = tr[:, tr_rg[r, 0]:tr_rg[r, 1]]
        h = kept.shape[1]
        assert blk.shape[1] == 2 * h and np.array_equal(blk[:, :h], kept) and np.array_equal(blk[:, h:], kept[::-1])
        assert (tr_et[tr_rg[r, 0]:tr_rg[r, 1]] == r).all()
        assert te_rg[r, 1] - te_rg[r, 0] == 2 * ((b - a) - h)
    assert tr.shape[1] + te.shape[1] == 2 * ptr[-1]
    # same Bernoulli(p) law as the reference's draw: |kept - n p| within 4 sigma for both, per relation
    ref = process_edges([torch.from_numpy(pairs[:, ptr[r]:ptr[r + 1]]) for r in range(len(sizes))], p=0.9,
                        rng=np.random.RandomState(5))
    for r, n in enumerate(sizes):
        sd = max(1.0, np.sqrt(n * 0.9 * 0.1))
        assert abs((tr_rg[r, 1] - tr_rg[r, 0]) / 2 - 0.9 * n) <= 4 * sd
        assert abs(int(ref[2][r, 1] - ref[2][r, 0]) / 2 - 0.9 * n) <= 4 * sd
    assert split_flags_spec(1000, 1.0, 3).all() and not split_flags_spec(1000, 0.0, 3).any()
    assert not np.array_equal(split_flags_spec(1000, 0.5, 3), split_flags_spec(1000, 0.5, 4))


def test_sym_adj_csr_ingest_matches_triu_pairs():
    """`sym_adj_to_pairs`: the reference's sym_adj layout (stacked symmetric CSR) -> the upper-triangular pair
    list of data/utils.py:60,151, without per-relation lists; checked against scipy."""
    import scipy.sparse as sp
    from tip_amd.data import sym_adj_to_pairs
    rng = np.random.RandomState(1)
    n, mats = 23, []
    for m in (40, 0, 7, 150):
        u, v = rng.randint(0, n, m), rng.randint(0, n, m)
        keep = u != v
        a = sp.coo_matrix((np.ones(keep.sum()), (u[keep], v[keep])), shape=(n, n)).tocsr()
        a = ((a + a.T) > 0).astype(np.float32).tocsr()
        a.sort_indices()
        mats.append(a)
    indptr = torch.from_numpy(np.stack([a.indptr for a in mats]).astype(np.int64))
    indices = torch.from_numpy(np.concatenate([a.indices for a in mats]).astype(np.int64))
    pairs, rel_ptr = sym_adj_to_pairs(indptr, indices, n)
    for r, a in enumerate(mats):
        coo = sp.triu(a).tocsr().tocoo()
        got = pairs[:, rel_ptr[r]:rel_ptr[r + 1]].numpy()
        assert np.array_equal(got, np.stack([coo.row, coo.col]))


def test_segmented_gather_plan_semantics_and_launch_order():
    """`build_gather_plan_segmented` (large D-D graphs): same sums as the default plan; items run
    segment by segment; every row's pieces get consecutive partial slots; rows without edges are written."""
    from tip_amd.plan import build_gather_plan_segmented, relations_per_segment
    g = torch.Generator().manual_seed(0)
    N, R, E, d = 57, 23, 4000, 4
    rel = torch.sort(torch.randint(0, R, (E,), generator=g)).values
    src = torch.randint(0, N, (E,), generator=g)
    dst = torch.randint(0, N - 4, (E,), generator=g)                     # the last 4 rows stay empty
    dst[:700] = 3                                                        # a hub row
    y = torch.randn(R * N, d, generator=g, dtype=torch.float64)
    want = torch.zeros(N, d, dtype=torch.float64).index_add_(0, dst, y[rel * N + src])
    for B, chunk in ((4, 16), (1, 8), (100, 32), (5, 1000)):
        plan = build_gather_plan_segmented(dst, rel * N + src, rel // B, N, R * N, chunk)
        torch.testing.assert_close(execute_plan_reference(plan, y), want)
        torch.testing.assert_close(execute_plan_reference(plan, y, torch.full((N,), 0.5, dtype=torch.float64)), want * 0.5)
        it = plan.items.long()
        nonempty = it[:, 1] > it[:, 0]
        segs = (plan.row_id.long()[it[nonempty, 0]] // N) // B
        assert bool((segs[1:] >= segs[:-1]).all())                       # segment-major launch order
        assert int((it[:, 1] - it[:, 0]).max()) <= chunk
        sr = plan.split_rows.long()
        assert int((sr[:, 2] - sr[:, 1]).sum()) == plan.n_slots and plan.max_slots == int((sr[:, 2] - sr[:, 1]).max())
        assert bool((sr[1:, 1] == sr[:-1, 2]).all())                     # slots tile [0, n_slots) row by row
    with pytest.raises(ValueError):                                      # edges not grouped by relation
        build_gather_plan_segmented(dst, rel * N + src, torch.flip(rel, [0]) // 4, N, R * N, 16)
    assert relations_per_segment(10000, 128) == 13 and relations_per_segment(10 ** 9, 128) == 1


def test_bench_roofline_helpers(tmp_path, monkeypatch):
    """bench.py's bookkeeping (no GPU): PMC summaries are only quoted when they carry the build id of the running library,
    the step floor prices every launch by its own bound, the roofline object of an MFMA kernel reports both flop counts."""
    import importlib
    import json as js
    bench = importlib.import_module('bench')
    prof = tmp_path / 'profiles'
    prof.mkdir()
    doc = {'build_id': 'aaaa', 'step_hbm_bytes': 9.0e8,
           'kernels': {'node_products_kernel<1> grid=294144x1x1': {'hbm_bytes_per_launch': 1.0e8}}}
    (prof / 'r09_pmc_traffic.json').write_text(js.dumps(doc))
    (prof / 'r09_kernel_by_grid.csv').write_text('# x\nkernel_and_grid,calls,avg_ns,min_ns,max_ns\n"node_products_kernel<1> grid=294144x1x1",5,40500,1,2\n')
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    assert bench.pmc_traffic('node_products_kernel', None, 'bbbb') == (None, None, None, None)    # other build: dropped
    t, us, src, _raw = bench.pmc_traffic('node_products_kernel', None, 'aaaa')
    assert t == 1.0e8 and abs(us - 40.5) < 1e-9 and src == 'r09_pmc_traffic.json'
    assert bench.step_hbm_bytes('bbbb') is None and bench.step_hbm_bytes('aaaa')[0] == 9.0e8
    # the run's own PMC passes: a child runs prepare() + warm-up + steps FULL-SIZE steps and nothing else;
    # bytes of one step = sum over libtipk's kernels of bytes per launch x launches / that count -- torch's kernels excluded --
    # and every figure exists corrected (2 x FETCH_SIZE + WRITE_SIZE) and uncorrected
    per = {'pair_grads_kernel<32> grid=165120x1x1': {'FETCH_SIZE': (10.0, 5), 'WRITE_SIZE': (20.0, 5)},
           'gather_sum_kernel<4> grid=1024x1x1': {'FETCH_SIZE': (1.0, 10), 'WRITE_SIZE': (0.5, 10)},
           'at::native::vectorized_elementwise_kernel<4> grid=64x1x1': {'FETCH_SIZE': (100.0, 50), 'WRITE_SIZE': (100.0, 50)}}
    live = bench.pmc_summary(per, 5)
    k = 'pair_grads_kernel<32> grid=165120x1x1'
    assert live['kernels'][k] == 40.0 * 1024 and live['kernels_uncorrected'][k] == 30.0 * 1024 and live['steps_run'] == 5
    assert live['step_hbm_bytes'] == (40.0 * 5 + 2.5 * 10) * 1024 / 5
    assert live['step_hbm_bytes'] == sum(live['kernels'][q] * n for q, n in ((k, 5), ('gather_sum_kernel<4> grid=1024x1x1', 10))) / 5
    assert live['step_hbm_bytes_uncorrected'] == (30.0 * 5 + 1.5 * 10) * 1024 / 5
    monkeypatch.setattr(bench, '_LIVE_PMC', live)
    t, us, src, raw = bench.pmc_traffic('pair_grads_kernel<32>', None, 'zzzz')
    assert (t, us, src, raw) == (40.0 * 1024, None, 'this run', 30.0 * 1024)
    assert bench.step_hbm_bytes('zzzz') == (live['step_hbm_bytes'], 'this run', live['step_hbm_bytes_uncorrected'])
    roof = bench.roofline_of({'label': 'pair_grads[dd.bwd,d=32]', 'key': 'pair_grads_kernel<32>', 'grid': None, 'bound': 'mfma',
                              'work': 5.2e8, 'note': 'n'}, 10.0, 'zzzz')
    assert roof['traffic'] == 40.0 * 1024 and roof['traffic_uncorrected'] == 30.0 * 1024 and roof['note'] == 'n'
    monkeypatch.setattr(bench, '_LIVE_PMC', None)
    rec = {'label': 'node_products[dd.bwd,d=32]', 'key': 'node_products_kernel', 'grid': None, 'bound': 'mfma', 'work': 1.35e9,
           'rows': 329188, 'flops_dense_form': 2.9e9}
    roof = bench.roofline_of(rec, 40.0, 'aaaa')
    assert roof['bound'] == 'mfma' and abs(roof['frac'] - 1.35e9 / 40e-6 / bench.MFMA_F32_PEAK) < 1e-12
    assert abs(roof['frac_dense_form'] - 2.9e9 / 40e-6 / bench.MFMA_F32_PEAK) < 1e-12 and roof['traffic'] == 1.0e8
    launches = [rec, {'label': 'rel_stream[dd.bwd,d=32]', 'key': 'k', 'grid': 'g', 'bound': 'lds', 'work': 1.1e9, 'aggregation': True}]
    kern = {'gather_sum[pp.fwd,d=32]': (3, 0.02), 'node_products[x]': (3, 0.04), 'rel_stream[y]': (3, 0.03), 'misc': (6, 0.005)}
    fl = bench.step_floor({}, launches, kern, 1_450_000, {})
    assert fl['launches_per_step'] == 5 and abs(fl['parts_us']['node_products[dd.bwd,d=32]'] - 1.35e9 / bench.MFMA_F32_PEAK * 1e6) < 0.01
    assert abs(fl['us'] - sum(fl['parts_us'].values())) < 0.05 and fl['parts_us']['2 other launches x 2.0 us'] == 4.0


def test_sampler_units_tile_the_positions_and_balance():
    """tip_amd.neg_sampling.sampler_units: the bitmap sampler's deal of (relation, range) units -- units tile [0, E),
    stay inside their relation, and no workgroup carries much more than the mean although one relation alone is larger
    than the mean load (BioSNAP: 51 466 positions against 32 525)."""
    from tip_amd.neg_sampling import sampler_units
    sizes = [51466, 48612, 0, 7, 300] + [8000 + 13 * i for i in range(900)] + [1]
    rel_ptr = np.r_[0, np.cumsum(sizes)]
    n_wg = 256
    ptr, units = sampler_units(torch.from_numpy(rel_ptr), n_wg)
    assert ptr.dtype == torch.int32 and units.dtype == torch.int32 and ptr.numel() == n_wg + 1 and int(ptr[-1]) == units.shape[0]
    u = sorted(units.tolist(), key=lambda x: x[1])
    assert u[0][1] == 0 and u[-1][2] == rel_ptr[-1] and all(u[i][2] == u[i + 1][1] for i in range(len(u) - 1))
    for r, a, b in u:
        assert rel_ptr[r] <= a < b <= rel_ptr[r + 1]
    loads = [sum(int(x[2] - x[1]) for x in units[ptr[w]:ptr[w + 1]].tolist()) for w in range(n_wg)]
    assert sum(loads) == rel_ptr[-1] and max(loads) < 1.15 * (sum(loads) / n_wg)
    assert sampler_units(torch.from_numpy(rel_ptr), n_wg)[1].tolist() == units.tolist()          # deterministic
    p0, u0 = sampler_units(torch.zeros(1, dtype=torch.int64), 4)
    assert p0.tolist() == [0] * 5 and u0.shape == (0, 3)


def test_stream_plan_with_8_byte_rows_interprets_correctly():
    """build_stream_plan_rows(row_bytes=8): the plan of the P-P graph's 2-column blocks (64 one-lane slots per wavefront),
    executed by the CPU interpreter of the record format == a plain index_add; every row written exactly once."""
    from tip_amd.plan import build_stream_plan_rows, execute_stream_plan_reference
    g = torch.Generator().manual_seed(12)
    N, E = 700, 9000
    dst, src = torch.randint(0, N, (E,), generator=g), torch.randint(0, N, (E,), generator=g)
    dst[:2500] = 3                                                     # hub row -> wide run
    dst = torch.where(dst == 5, torch.full_like(dst, 6), dst)          # row without edges
    sp = build_stream_plan_rows(dst, src, N, N, 2, 1, row_bytes=8)
    assert sp.row_bytes == 8 and sp.lanes == 1 and sp.cells.shape[1] == 64 and sp.idx_unit in (1, 2, 4, 8)
    x = torch.randn(N, 2, generator=g)
    got = execute_stream_plan_reference(sp, x)
    want = torch.zeros(N, 2).index_add_(0, dst, x[src])
    assert torch.allclose(got, want, atol=1e-4)
    assert 5 in sp.zero_rows.tolist()


def test_adam_adopts_reloaded_step_counts():
    """after `load_state_dict` (torch casts a capturable optimizer's `step` to float32) and for checkpoints
    written by torch.optim.Adam, `step` is re-made as an int64 counter -- host logic only, no kernel launch."""
    from tip_amd.optim import Adam
    p = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(4))]
    ref = torch.optim.Adam(p, lr=0.01)
    for q in p:
        q.grad = torch.ones_like(q)
    ref.step(); ref.step()
    mine = Adam([torch.nn.Parameter(q.detach().clone()) for q in p], lr=0.01)
    mine.load_state_dict(ref.state_dict())
    for q in mine.param_groups[0]['params']:
        st = mine.state[q]
        assert st['step'].dtype == torch.int64 and st['step'].dim() == 0 and int(st['step']) == 2
        assert st['exp_avg'].dtype == torch.float32 and st['exp_avg'].shape == q.shape
    again = Adam([torch.nn.Parameter(q.detach().clone()) for q in p], lr=0.01)
    again.load_state_dict(mine.state_dict())                           # own checkpoint: torch's cast to float32 is undone
    assert all(again.state[q]['step'].dtype == torch.int64 and int(again.state[q]['step']) == 2
               for q in again.param_groups[0]['params'])
    assert all(g['capturable'] for g in again.param_groups)


def test_gcn_rows_subset_plans_and_compact_hierarchy_graph():
    """conv2 of the P-P encoder only for the rows the P -> D stage reads.  The restricted plans give
    exactly the full graph's rows (forward) and the full graph's transposed aggregate of a gradient that is zero outside the
    rows (backward); the P -> D graph over the compact source block equals the one over the whole block."""
    from tip_amd.layers import gcn_norm_graph, hier_graph
    g = torch.Generator().manual_seed(21)
    N, E, d = 57, 400, 4
    ei = torch.randint(0, N, (2, E), generator=g)
    ei[:, :5] = torch.arange(5)                                      # self loops (replaced by unit loops)
    rows = torch.tensor(sorted({3, 4, 9, 20, 21, 50, 56}))
    full = gcn_norm_graph(ei, N, d=d)
    sub = gcn_norm_graph(ei, N, d=d, rows=rows)
    x = torch.randn(N, d, generator=g, dtype=torch.float64)
    want = execute_plan_reference(full.fwd, x)
    got = execute_plan_reference(sub.fwd, x)
    assert sub.fwd.n_out == rows.numel() and sub.fwd.n_table == N and sub.bwd.n_out == N and sub.bwd.n_table == rows.numel()
    torch.testing.assert_close(got, want[rows], rtol=1e-12, atol=1e-12)
    gc = torch.randn(rows.numel(), d, generator=g, dtype=torch.float64)
    gf = torch.zeros(N, d, dtype=torch.float64)
    gf[rows] = gc
    torch.testing.assert_close(execute_plan_reference(sub.bwd, gc), execute_plan_reference(full.bwd, gf), rtol=1e-12, atol=1e-12)
    # P -> D over the compact block
    n_t = 11
    dp = torch.stack([rows[torch.randint(0, rows.numel(), (40,), generator=g)], N + torch.randint(0, n_t - 2, (40,), generator=g)])
    hg = hier_graph(dp, N + n_t, N, table_rows=N, d=d)
    inv = torch.full((N,), -1, dtype=torch.int64)
    inv[rows] = torch.arange(rows.numel())
    hc = hier_graph(torch.stack([inv[dp[0]], dp[1] - N + rows.numel()]), rows.numel() + n_t, rows.numel(), table_rows=rows.numel(), d=d)
    h = torch.randn(N, d, generator=g, dtype=torch.float64)
    torch.testing.assert_close(execute_plan_reference(hc.fwd, h[rows], hc.scale.double()),
                               execute_plan_reference(hg.fwd, h, hg.scale.double()), rtol=1e-12, atol=1e-12)
    gm = torch.randn(n_t, d, generator=g, dtype=torch.float64)
    torch.testing.assert_close(execute_plan_reference(hc.bwd, gm), execute_plan_reference(hg.bwd, gm)[rows], rtol=1e-12, atol=1e-12)
    assert float(execute_plan_reference(hg.bwd, gm).abs().sum() - execute_plan_reference(hg.bwd, gm)[rows].abs().sum()) == 0.0


@pytest.mark.parametrize('R,N,E', [(70, 23, 900), (5, 16, 40), (33, 9, 0)])
def test_row_stream_plan_reference(R, N, E):
    """`build_row_stream_plan` (include/tipk.h section 2h) interpreted in torch as the kernel walks it (batches of 16 words
    per lane half, padding into the dump column) == the definition; every edge appears once, in its (node, tile, half)
    list, rows in edge-list order."""
    from tip_amd.plan import build_row_stream_plan, execute_row_stream_reference
    g = torch.Generator().manual_seed(R + N)
    rel = torch.randint(0, R, (E,), generator=g)
    key = torch.randint(0, N, (E,), generator=g)
    other = torch.randint(0, N, (E,), generator=g)
    rp = build_row_stream_plan(key, other, rel, N, R)
    assert rp.desc.shape == (N, (R + 31) // 32, 2) and rp.entries.dtype == torch.int32
    words = rp.entries.view(-1)
    real = words[words != 128].to(torch.int64)
    assert real.numel() == E
    nb, ch = 4, 8
    table, att, xb = torch.randn(N, ch, generator=g), torch.randn(R, nb, generator=g), torch.randn(nb, N, ch, generator=g)
    s = torch.zeros(R * N, ch, dtype=torch.float64)
    if E:
        s.index_add_(0, rel * N + key, table.double()[other])
    s = s.view(R, N, ch)
    t, datt = execute_row_stream_reference(rp, table, att, xb)
    assert torch.allclose(t, torch.einsum('rb,rvc->bvc', att.double(), s), rtol=1e-12, atol=1e-12)
    assert torch.allclose(datt, torch.einsum('rvc,bvc->rb', s, xb.double()), rtol=1e-12, atol=1e-12)
    # `inside` (byte 3) is set exactly where the previous word of the half's list has the same row -- the kernel's running
    # sum restarts at every other word; the halves of a (node, tile) hold disjoint rows and are balanced up to one row
    w = rp.entries.to(torch.int64)
    desc = rp.desc.to(torch.int64)
    for v in range(N):
        for tl in range(rp.n_tiles):
            f, nb_ = int(desc[v, tl, 0]), int(desc[v, tl, 1])
            seen = []
            for h in range(2):
                lst = w[f:f + nb_, h, :].reshape(-1)
                real = lst[lst != 128]
                assert bool((lst[real.numel():] == 128).all())                 # padding only at the end of a half's list
                rows = real & 0xff
                assert bool((rows[1:] >= rows[:-1]).all())
                want_inside = torch.zeros_like(rows)
                want_inside[1:] = (rows[1:] == rows[:-1]).to(torch.int64)
                assert torch.equal(real >> 24, want_inside)
                seen.append(rows)
            assert not (set(seen[0].tolist()) & set(seen[1].tolist()))
            n0, n1 = seen[0].numel(), seen[1].numel()
            assert nb_ >= 1
            if n0 + n1:
                assert nb_ == max(1, (max(n0, n1) + 15) // 16)
                longest_row = int(torch.bincount(torch.cat(seen) // 4).max())
                assert abs(n0 - n1) <= longest_row


def test_pair_link_words_bit_layout():
    """`layers.pair_link_words` (tipk.h section 2c `links`): bit r of word (u, t) = some edge u -> 32 t + r; rows padded to a
    multiple of 8 nodes, columns to whole words; bit 31 lands in the sign of the int32 the words travel as."""
    from tip_amd.layers import pair_link_words
    g = torch.Generator().manual_seed(3)
    n = 77
    src, dst = torch.randint(0, n, (400,), generator=g), torch.randint(0, n, (400,), generator=g)
    src = torch.cat([src, torch.tensor([5, 5])]); dst = torch.cat([dst, torch.tensor([31, 63])])      # sign bits of two words
    w = pair_link_words(src, dst, n)
    assert w.dtype == torch.int32 and tuple(w.shape) == (80, 3)
    dense = torch.zeros(80, 96, dtype=torch.bool)
    dense[src, dst] = True
    bits = ((w.to(torch.int64) & 0xffffffff).unsqueeze(-1) >> torch.arange(32)) & 1
    assert torch.equal(bits.view(80, 96).bool(), dense)
    assert int(w[5, 0]) < 0 and int(w[5, 1]) < 0


@pytest.mark.parametrize('R,N,E', [(70, 23, 900), (5, 16, 40), (33, 9, 0)])
def test_row_stream_plan_s_reference(R, N, E):
    """`build_row_stream_plan_s` (wave-uniform entries of `tipk_rgcn_row_products_s`) interpreted in torch == the definition;
    one sorted list per (node, tile), `inside` exactly where the previous entry has the same row, padding only at the end."""
    from tip_amd.plan import build_row_stream_plan_s, execute_row_stream_s_reference, RowStreamPlanS
    g = torch.Generator().manual_seed(R + N)
    rel = torch.randint(0, R, (E,), generator=g)
    key = torch.randint(0, N, (E,), generator=g)
    other = torch.randint(0, N, (E,), generator=g)
    rp = build_row_stream_plan_s(key, other, rel, N, R)
    assert rp.desc.shape == (N, (R + 31) // 32, 2) and rp.entries.dtype == torch.int32
    nb, ch = 4, 8
    table, att, xb = torch.randn(N, ch, generator=g), torch.randn(R, nb, generator=g), torch.randn(nb, N, ch, generator=g)
    s = torch.zeros(R * N, ch, dtype=torch.float64)
    if E:
        s.index_add_(0, rel * N + key, table.double()[other])
    s = s.view(R, N, ch)
    t, datt = execute_row_stream_s_reference(rp, table, att, xb)
    assert torch.allclose(t, torch.einsum('rb,rvc->bvc', att.double(), s), rtol=1e-12, atol=1e-12)
    assert torch.allclose(datt, torch.einsum('rvc,bvc->rb', s, xb.double()), rtol=1e-12, atol=1e-12)
    w = rp.entries.to(torch.int64)
    desc = rp.desc.to(torch.int64)
    pad = 32 * RowStreamPlanS.ROW_BYTES
    n_real = 0
    for v in range(N):
        for tl in range(rp.n_tiles):
            f, nb_ = int(desc[v, tl, 0]), int(desc[v, tl, 1])
            assert nb_ >= 1
            w1 = w[f:f + nb_, 1, :].reshape(-1)
            real = w1[w1 != pad]
            assert bool((w1[real.numel():] == pad).all()) and nb_ == max(1, (real.numel() + 15) // 16)
            rows = real & 0xffff
            assert bool((rows % RowStreamPlanS.ROW_BYTES == 0).all()) and bool((rows[1:] >= rows[:-1]).all())
            want_inside = torch.zeros_like(rows)
            want_inside[1:] = (rows[1:] == rows[:-1]).to(torch.int64) * 0x3f800000
            assert torch.equal(real & 0x3f800000, want_inside)
            n_real += real.numel()
    assert n_real == E
    e512 = rp.entries_for(512)
    assert torch.equal(e512[:, 0, :], rp.entries[:, 0, :] * 512) and torch.equal(e512[:, 1, :], rp.entries[:, 1, :])


@pytest.mark.parametrize('symmetric,part_rows', [(True, 40), (False, 64), (True, 1008)])
def test_pair_bwd_plan_reference(symmetric, part_rows):
    """`build_pair_bwd_plan` (include/tipk.h section 2e), interpreted in torch exactly as the two kernels walk it: slots,
    tiles, cell lines (half the cells on a symmetric graph), partitions of the (symmetrised) pair-gradient table, the merged
    wave-stream plan (every (partition, relation) row written once, by a wavefront of that partition) == the definition."""
    from tip_amd.plan import build_pair_bwd_plan, execute_pair_bwd_reference
    g = torch.Generator().manual_seed(7 + part_rows)
    N, R, nb, d = 41, 13, 32, 16
    src, dst, rel = [], [], []
    for r in range(R):
        m = 1 if r == 3 else int(torch.randint(2, 70, (1,), generator=g))
        u, v = torch.randint(0, 30, (m,), generator=g), torch.randint(0, 30, (m,), generator=g)
        k = u != v
        u, v = u[k], v[k]
        if symmetric:
            key = torch.unique(torch.minimum(u, v) * N + torch.maximum(u, v))
            u, v = torch.cat([key // N, key % N]), torch.cat([key % N, key // N])
        else:
            key = torch.unique(u * N + v)
            u, v = key // N, key % N
        src.append(u); dst.append(v); rel.append(torch.full((u.numel(),), r))
    src, dst, rel = torch.cat(src), torch.cat(dst), torch.cat(rel)
    scale = (1.0 / torch.bincount(dst, minlength=N).clamp(min=1).double()).float()
    plan = build_pair_bwd_plan(src, dst, rel, N, R, scale, symmetric, n_wg=6, part_rows_max=part_rows)
    assert plan.n_slots % 32 == 0 and plan.part_first.shape == (plan.n_parts,) and int(plan.part_first[0]) == 0
    assert int(plan.node_desc[:, 2].sum()) * 32 == plan.n_slots and bool((plan.node_desc[:-1, 2] >= plan.node_desc[1:, 2]).all())
    att = torch.randn(R, nb, generator=g, dtype=torch.float64)
    xb = torch.randn(N, nb, d, generator=g, dtype=torch.float64)
    gz = torch.randn(N, d, generator=g, dtype=torch.float64)
    C = torch.zeros(N * N, nb, dtype=torch.float64)
    C.index_add_(0, src * N + dst, att[rel])
    cells = C.clone().view(N, N, nb)
    if symmetric:
        cells[~torch.triu(torch.ones(N, N, dtype=torch.bool))] = float('nan')        # never read
    dxb, pg, datt = execute_pair_bwd_reference(plan, cells.view(N * N, nb), xb, gz, nb)
    gp = gz * scale.double().unsqueeze(1)
    want_xb = torch.einsum('uvb,vc->buc', C.view(N, N, nb), gp)
    want_att = torch.zeros(R, nb, dtype=torch.float64)
    want_att.index_add_(0, rel, torch.einsum('ebc,ec->eb', xb[src], gp[dst]))
    torch.testing.assert_close(dxb, want_xb, rtol=1e-12, atol=1e-12)
    torch.testing.assert_close(datt, want_att, rtol=1e-12, atol=1e-12)
    assert not bool(torch.isnan(pg).any())


def test_row_plan_support_is_asked_before_the_lazy_plan_is_built():
    """with 65 536 < N < 2^23 and a width that is no multiple of 64 the backward pass evaluated the lazy
    `graph.row_bwd` (-> build_row_stream_plan asserts N <= 65 536) before asking the kernel whether it takes the shape.
    `_row_plan` asks first; a shape the kernels do not take never touches the builders."""
    from tip_amd import ops

    def boom():
        raise AssertionError('the lazy plan was built for a shape the kernel does not take')
    big = ops.AggGraph(None, None, row_fwd=boom, row_bwd=boom, row_wave_uniform=False)
    assert ops._row_plan(big, True, 70000, 100, 32, 32) is None and ops._row_plan(big, False, 70000, 100, 32, 96) is None
    wave = ops.AggGraph(None, None, row_fwd=boom, row_bwd=boom, row_wave_uniform=True)
    assert ops._row_plan(wave, True, 70000, 100, 32, 96) is None           # width no multiple of 64
    assert ops._row_plan(wave, True, 5_000_000, 100, 32, 128) is None      # table beyond 2 GB of 32-bit byte offsets
    none = ops.AggGraph(None, None)
    assert ops._row_plan(none, False, 1000, 10, 32, 64) is None
    built = []
    ok = ops.AggGraph(None, None, row_fwd=lambda: built.append('f') or 'F', row_bwd=lambda: built.append('b') or 'B', row_wave_uniform=True)
    assert ops._row_plan(ok, False, 70000, 100, 32, 128) == 'F' and ops._row_plan(ok, True, 70000, 100, 32, 64) == 'B' and built == ['f', 'b']


def test_pd_stage_row_deal_and_forward_workgroup_descs():
    """Round 6 plans of the fused P -> D launches (tip_amd/layers.py): source rows dealt to workgroups by EDGES (a cluster of hub
    rows must not land in one workgroup), and the forward launch's rows dealt by edge count with W wavefronts per row."""
    from tip_amd.layers import deal_rows_by_edges, drug_workgroups
    g = torch.Generator().manual_seed(3)
    counts = torch.randint(0, 9, (500,), generator=g).tolist()
    counts[200:240] = [90] * 40                                        # a cluster of hubs, as in BioSNAP's P -> D graph
    counts[300] = 2000                                                  # one row beyond the edge limit: alone
    b = deal_rows_by_edges(counts, 64, 512)
    assert b[0] == 0 and b[-1] == 500 and all(x < y for x, y in zip(b, b[1:]))
    for lo, hi in zip(b, b[1:]):
        e = sum(counts[lo:hi])
        assert hi - lo <= 64 and (e <= 512 or hi - lo == 1), (lo, hi, e)
    assert max(sum(counts[lo:hi]) for lo, hi in zip(b, b[1:]) if hi - lo > 1) <= 512
    assert deal_rows_by_edges([], 64, 512) == [0, 0] and deal_rows_by_edges([5], 64, 512) == [0, 1]
    cnt = torch.tensor([3, 700, 0, 65, 512, 64, 1, 100, 99, 98, 97] + [2] * 40)
    order, descs = drug_workgroups(cnt)
    seen = []
    for first, packed in descs:
        n, w = packed & 255, packed >> 8
        rows = order[first:first + n].tolist()
        seen += rows
        assert n >= 1 and n * w <= 16 and w in (1, 4, 16)
        for r in rows:
            c = int(cnt[r])
            assert (w == 16) == (c > 512) and (w == 4) == (64 < c <= 512)
    assert sorted(seen) == list(range(cnt.numel()))
