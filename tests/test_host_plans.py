"""The C++ plan builders of libtipk (include/tipk.h section 10c, tip_amd/csrc/tipk_pairplan.hip) against tip_amd/plan.py, array
by array, bit for bit -- host code only, runs without a GPU.  The graph handle of the op-level C ABI builds its pair-form plans
with the C++ side; the PyTorch modules with the Python side; the kernels are the same."""
import ctypes as C

import numpy as np
import pytest
import torch

from tip_amd import _lib
from tip_amd.layers import pair_link_words
from tip_amd.plan import build_gather_plan, build_pair_bwd_plan, build_stream_plan_rows, group_slots_for


def _i64(t):
    a = np.ascontiguousarray(t.numpy().astype(np.int64))
    return a, a.ctypes.data_as(C.c_void_p)


class HostPlan(object):
    def __init__(self, handle):
        self.h, self.lib = handle, _lib.lib()

    def array(self, name):
        data, count, eb = C.c_void_p(), C.c_int64(), C.c_int()
        assert self.lib.tipk_host_plan_array(self.h, name.encode(), C.byref(data), C.byref(count), C.byref(eb)) == 0, name
        dt, ct = {2: (np.uint16, C.c_uint16), 4: (np.int32, C.c_int32), 8: (np.int64, C.c_int64)}[eb.value]
        if count.value == 0:
            return np.zeros(0, dtype=dt)
        return np.ctypeslib.as_array(C.cast(data, C.POINTER(ct)), shape=(count.value,)).astype(dt).copy()

    def scalar(self, name):
        return int(self.lib.tipk_host_plan_scalar(self.h, name.encode()))

    def free(self):
        self.lib.tipk_host_plan_free(self.h)


def _same_stream_plan(hp, sp, pre=''):
    for name in ('n_rows', 'n_table', 'n_bands', 'n_edges', 'n_wg', 'lanes', 'piece', 'idx_unit', 'row_bytes'):
        assert hp.scalar(pre + name) == int(getattr(sp, name)), name
    assert np.array_equal(hp.array(pre + 'wave_ptr'), sp.wave_ptr.numpy())
    assert np.array_equal(hp.array(pre + 'cells'), sp.cells.numpy().reshape(-1))
    assert np.array_equal(hp.array(pre + 'ids'), sp.ids.numpy().view(np.uint16).reshape(-1))
    assert np.array_equal(hp.array(pre + 'zero_ptr'), sp.zero_ptr.numpy())
    assert np.array_equal(hp.array(pre + 'zero_rows'), sp.zero_rows.numpy())


def _random_runs(gen, n_rows, n_table, n_edges, hubs):
    """Output rows with a heavy tail (some rows with hundreds of edges: the WIDE runs), some rows empty."""
    w = torch.rand(n_rows, generator=gen) ** 4
    w[torch.randperm(n_rows, generator=gen)[:n_rows // 5]] = 0          # rows without edges
    for h in range(hubs):
        w[int(torch.randint(0, n_rows, (1,), generator=gen))] = 30.0 * (h + 1)
    out_row = torch.multinomial(w / w.sum(), n_edges, replacement=True, generator=gen)
    tab_row = torch.randint(0, n_table, (n_edges,), generator=gen)
    return out_row, tab_row


@pytest.mark.parametrize('n_rows,n_table,n_edges,n_wg,lanes,piece,hubs', [
    (300, 97, 5000, 4, 8, 4, 3),          # pair cells of a small graph: 128-byte rows
    (2000, 1097, 60000, 16, 8, 4, 6),     # BioSNAP-like table height, wide runs of k = 2, 4, 8
    (500, 2000, 20000, 8, 4, 4, 4),       # 64-byte rows (a table split into two column blocks)
    (64, 33, 900, 2, 16, 4, 1),           # 256-byte rows: no bank classes
    (128, 50, 0, 2, 8, 4, 0),             # no edges at all
    (40, 7, 37, 1, 2, 2, 0),              # 32-byte rows, short pieces
])
def test_stream_plan_rows_matches_plan_py(n_rows, n_table, n_edges, n_wg, lanes, piece, hubs):
    gen = torch.Generator().manual_seed(n_rows * 7 + lanes)
    if n_edges:
        out_row, tab_row = _random_runs(gen, n_rows, n_table, n_edges, hubs)
    else:
        out_row = tab_row = torch.zeros(0, dtype=torch.int64)
    sp = build_stream_plan_rows(out_row, tab_row, n_rows, n_table, n_wg, lanes, piece)
    (a, pa), (b, pb) = _i64(out_row), _i64(tab_row)
    h = C.c_void_p()
    assert _lib.lib().tipk_plan_stream_rows(pa, pb, n_edges, n_rows, n_table, n_wg, lanes, piece, 0, 0, C.byref(h)) == 0
    hp = HostPlan(h)
    try:
        _same_stream_plan(hp, sp)
    finally:
        hp.free()


def test_stream_plan_rows_eight_byte_rows():
    """The P-P graph's form: lanes = 1, 8-byte rows in 32 bank-pair classes."""
    gen = torch.Generator().manual_seed(5)
    out_row, tab_row = _random_runs(gen, 700, 3000, 9000, 2)
    sp = build_stream_plan_rows(out_row, tab_row, 700, 3000, 4, 1, 4, row_bytes=8)
    (a, pa), (b, pb) = _i64(out_row), _i64(tab_row)
    h = C.c_void_p()
    assert _lib.lib().tipk_plan_stream_rows(pa, pb, 9000, 700, 3000, 4, 1, 4, 0, 8, C.byref(h)) == 0
    hp = HostPlan(h)
    try:
        _same_stream_plan(hp, sp)
    finally:
        hp.free()


def _dd_graph(gen, n, r, symmetric, density=0.25, big=None):
    """A relation-typed D-D edge list in the reference's order (per relation: u < v half, then the mirrored half), with
    duplicate edges inside a relation, self pairs, a one-edge relation and isolated nodes."""
    src, dst, rel = [], [], []
    live = torch.randperm(n, generator=gen)[:max(2, n - 3)]            # a few isolated nodes
    for k in range(r):
        m = 1 if k == 1 else int(torch.randint(2, big or max(3, int(density * n)), (1,), generator=gen))
        u = live[torch.randint(0, live.numel(), (m,), generator=gen)]
        v = live[torch.randint(0, live.numel(), (m,), generator=gen)]
        if symmetric:
            lo, hi = torch.minimum(u, v), torch.maximum(u, v)
            if k % 3 == 0 and m > 2:
                lo[1], hi[1] = lo[0], hi[0]                            # a duplicate edge inside the relation
            s, d = torch.cat([lo, hi]), torch.cat([hi, lo])
        else:
            s, d = u, v
        src.append(s); dst.append(d); rel.append(torch.full_like(s, k))
    return torch.cat(src), torch.cat(dst), torch.cat(rel)


@pytest.mark.parametrize('n,r,symmetric,n_wg,lanes', [
    (60, 12, True, 8, 8),
    (60, 12, False, 8, 8),
    (200, 300, True, 64, 8),              # several partitions, several workgroups each
    (90, 40, True, 16, 4),
    (33, 5, False, 4, 4),
])
def test_pair_bwd_plan_matches_plan_py(n, r, symmetric, n_wg, lanes):
    gen = torch.Generator().manual_seed(n + r)
    src, dst, rel = _dd_graph(gen, n, r, symmetric, big=4000 if n == 200 else None)
    deg = torch.bincount(dst, minlength=n).clamp(min=1).to(torch.float32)
    scale = (1.0 / deg).contiguous()
    pb = build_pair_bwd_plan(src, dst, rel, n, r, scale, symmetric, n_wg, lanes, 4)
    (a, pa), (b, pb_), (c, pc) = _i64(src), _i64(dst), _i64(rel)
    sc = np.ascontiguousarray(scale.numpy())
    h = C.c_void_p()
    assert _lib.lib().tipk_plan_pair_bwd(pa, pb_, pc, src.numel(), n, r, sc.ctypes.data_as(C.c_void_p), int(symmetric), n_wg, lanes, 4,
                                         C.byref(h)) == 0
    hp = HostPlan(h)
    try:
        for name in ('n_slots', 'n_parts', 'part_len', 'n_alloc'):
            assert hp.scalar(name) == int(getattr(pb, name)), name
        assert hp.scalar('symmetric') == int(symmetric)
        assert np.array_equal(hp.array('slots'), pb.slots.numpy().reshape(-1))
        assert np.array_equal(hp.array('node_desc'), pb.node_desc.numpy().reshape(-1))
        assert np.array_equal(hp.array('tile_node'), pb.tile_node.numpy())
        assert np.array_equal(hp.array('part_first'), pb.part_first.numpy())
        assert np.array_equal(hp.array('wg_part'), pb.wg_part.numpy())
        _same_stream_plan(hp, pb.gather, 'gather.')
        if n == 200:
            assert pb.n_parts > 1 and pb.gather.n_wg > pb.n_parts
    finally:
        hp.free()


def test_pair_bwd_plan_refuses_a_graph_that_is_not_symmetric():
    src, dst, rel = torch.tensor([0, 1, 3]), torch.tensor([1, 0, 2]), torch.tensor([0, 0, 0])      # (3, 2) without (2, 3)
    (a, pa), (b, pb_), (c, pc) = _i64(src), _i64(dst), _i64(rel)
    sc = np.ones(4, dtype=np.float32)
    h = C.c_void_p()
    assert _lib.lib().tipk_plan_pair_bwd(pa, pb_, pc, 3, 4, 1, sc.ctypes.data_as(C.c_void_p), 1, 4, 8, 4, C.byref(h)) == -2          # TIPK_EUNSUPPORTED
    assert not h.value


def test_link_words_match_layers_py():
    gen = torch.Generator().manual_seed(3)
    for n in (31, 64, 645):
        src, dst = torch.randint(0, n, (5 * n,), generator=gen), torch.randint(0, n, (5 * n,), generator=gen)
        want = pair_link_words(src, dst, n).numpy().reshape(-1)
        (a, pa), (b, pb_) = _i64(src), _i64(dst)
        h = C.c_void_p()
        assert _lib.lib().tipk_plan_link_words(pa, pb_, src.numel(), n, C.byref(h)) == 0
        hp = HostPlan(h)
        try:
            assert np.array_equal(hp.array('links'), want)
        finally:
            hp.free()


@pytest.mark.parametrize('n_out,n_table,n_edges,d,weights,hubs', [
    (500, 700, 20000, 32, True, 3),           # a GCN-like graph: weighted edges, a few hub rows cut into pieces
    (645, 19081 + 645, 18596, 16, True, 2),   # the P -> D stage's shape
    (64, 64 * 7, 60000, 32, False, 4),        # few destination rows with thousands of edges each (the D-D forward pass)
    (300, 300, 0, 16, False, 0),              # no edges: every row still gets its (empty) item
    (100, 50, 3000, 64, True, 1),             # d = 64: blocks of 64 slots
])
def test_grouped_gather_plan_matches_plan_py(n_out, n_table, n_edges, d, weights, hubs):
    gen = torch.Generator().manual_seed(n_out + d)
    if n_edges:
        out_row, tab_row = _random_runs(gen, n_out, n_table, n_edges, hubs)
    else:
        out_row = tab_row = torch.zeros(0, dtype=torch.int64)
    w = torch.rand(n_edges, generator=gen) if weights else None
    G = group_slots_for(d)
    gp = build_gather_plan(out_row, tab_row, n_out, n_table, edge_w=w, group_slots=G)
    (a, pa), (b, pb) = _i64(out_row), _i64(tab_row)
    wn = None if w is None else np.ascontiguousarray(w.numpy())
    h = C.c_void_p()
    assert _lib.lib().tipk_plan_gather(pa, pb, None if wn is None else wn.ctypes.data_as(C.c_void_p), n_edges, n_out, n_table, 0, G,
                                       C.byref(h)) == 0
    hp = HostPlan(h)
    try:
        assert hp.scalar('chunk') == gp.chunk and hp.scalar('group_slots') == G and hp.scalar('n_items') == gp.items.shape[0]
        assert np.array_equal(hp.array('items'), gp.items.numpy().reshape(-1))
        assert np.array_equal(hp.array('row_id'), gp.row_id.numpy())
        assert np.array_equal(hp.array('perm'), gp.perm.numpy())
        if weights and n_edges:
            assert np.array_equal(hp.array('edge_w'), gp.edge_w.numpy().view(np.int32))
        if hubs:
            assert int((gp.items[:, 3] & 4).ne(0).sum()) > 0                # some row was split
    finally:
        hp.free()
