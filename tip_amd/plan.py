"""Gather-sum plans: the static preprocessing of a graph for `tipk_gather_sum`.

The reference re-derives everything from COO `edge_index` on every call (PyG `propagate`:
`index_select` + `scatter`).  The graphs of the TIP path never change during training, so -- like
`GCNConv(cached=True)` at `src/layers.py:386-387` -- the work is done once:

  * edges are stably sorted by OUTPUT row (destination for a forward pass, source row for the
    transposed/backward pass), int64 ids are narrowed to int32;
  * every row's edge list is cut into work items of <= `chunk` edges (skewed rows -- a drug has
    up to 70 715 in-edges, SURVEY.md section 0 -- become many equal items; short rows one item);
  * items are ordered by decreasing length so the slots of one wavefront finish together;
  * rows that were split get consecutive slots in a `partial` buffer, summed in slot order by
    `tipk_gather_sum_finalize` -> results do not depend on scheduling (no float atomics);
  * with `group_slots` = G the pieces of a split row are instead placed in consecutive slots of ONE
    workgroup (blocks of G slots, best-fit packed, padded with null items) and added in order through
    LDS by the gather kernel itself: no partial buffer and no second launch (P-P and P->D graphs,
    where half of the rows are split and the finalize launch costs as much as a third of the gather).

Everything here is index arithmetic in torch (runs on CPU or GPU; unit-tested on CPU).  The layout
of `items` / `split_rows` is the contract of include/tipk.h section 1.
"""

import torch

DEFAULT_CHUNK = None          # None = `auto_chunk` (edges per work item chosen from the edge count)
TARGET_ITEMS = 65536          # ~ 256 CUs x 32 waves x 8 slots: one item per slot fills the chip


def auto_chunk(n_edges):
    """Power-of-two chunk in [16, 128] that yields about TARGET_ITEMS work items: short chains
    (few dependent index->row round trips per slot) while every CU still has a full set of waves.
    Graphs of up to ~4 M edges (P-P: 1.45 M) are latency-bound, not bandwidth-bound, and run 10 %
    faster with twice the items (measured: chunk 16 vs 32 on the BioSNAP P-P graph)."""
    target = TARGET_ITEMS * 2 if n_edges <= (1 << 22) else TARGET_ITEMS
    c = 16
    while c < 128 and n_edges // c > target:
        c *= 2
    return c


ITEM_DIRECT, ITEM_PIECE, ITEM_LEADER, ITEM_NULL = 1, 2, 4, 8      # items[:, 3] bits; leader: pieces << 8


class GatherPlan(object):
    """Device-resident plan.  Fields (all int32 unless noted):
    row_id [E]        source-table row per edge, in plan (sorted) order
    edge_w [E] fp32   optional per-edge weight in plan order
    items [n_items,4] (begin, end, target, flags)
    split_rows [m,3]  (out_row, first_slot, end_slot)
    perm [E] int64    plan order -> caller's edge order (for re-weighting)
    group_slots       0, or the block size G of the in-workgroup combination (see module doc)
    """

    def __init__(self, n_out, n_table, row_id, edge_w, items, split_rows, n_slots, perm, chunk, tag='',
                 group_slots=0):
        self.tag = tag
        self.n_out, self.n_table = int(n_out), int(n_table)
        self.row_id, self.edge_w = row_id, edge_w
        self.items, self.split_rows = items, split_rows
        self.n_slots, self.perm, self.chunk = int(n_slots), perm, int(chunk)
        self.group_slots = int(group_slots)
        self.n_edges = int(row_id.numel())
        self.max_slots = int((split_rows[:, 2] - split_rows[:, 1]).max()) if split_rows.shape[0] else 0
        self.seg_item_ptr = None      # segment-major plans: item range of every segment (host list)

    def to(self, device):
        mv = lambda t: None if t is None else t.to(device)
        p = GatherPlan(self.n_out, self.n_table, mv(self.row_id), mv(self.edge_w), mv(self.items),
                       mv(self.split_rows), self.n_slots, mv(self.perm), self.chunk, self.tag, self.group_slots)
        p.seg_item_ptr = self.seg_item_ptr
        return p

    @property
    def device(self):
        return self.items.device


def group_slots_for(d):
    """Block size G for plans whose rows are d floats wide: one workgroup of G * L threads runs one
    block (L lanes per item = next power of two of d/4 for vectorised rows, of d otherwise), at most
    1024 threads and at least one wavefront.  G = 128 for d <= 32 (hub rows of up to 128 pieces)."""
    lanes = 1
    need = d // 4 if d % 4 == 0 else d
    while lanes < need:
        lanes *= 2
    cap = 128                                                          # (swept in round 2)
    return max(min(cap, 1024 // lanes), 64 // lanes)


def pack_blocks(pieces, cap):
    """Best-fit-decreasing packing of rows with `pieces[i]` (<= cap) consecutive slots into blocks of
    `cap` slots.  -> (block, offset) per row, number of blocks.  Deterministic."""
    order = sorted(range(len(pieces)), key=lambda i: (-pieces[i], i))
    free = [[] for _ in range(cap + 1)]             # free[c] = blocks with c slots left (stack)
    used = []                                       # slots used per block
    block = [0] * len(pieces)
    offset = [0] * len(pieces)
    for i in order:
        p = pieces[i]
        b = -1
        for c in range(p, cap + 1):
            if free[c]:
                b = free[c].pop()
                break
        if b < 0:
            b = len(used)
            used.append(0)
        block[i], offset[i] = b, used[b]
        used[b] += p
        if cap - used[b] > 0:
            free[cap - used[b]].append(b)
    return block, offset, len(used)


def build_gather_plan(out_row, table_row, n_out, n_table, edge_w=None, chunk=DEFAULT_CHUNK, tag='', group_slots=0):
    """Plan for  out[o] = sum_{e: out_row[e]=o} edge_w[e] * table[table_row[e]].

    out_row, table_row: int64 [E] (any device); n_out / n_table: row counts of `out` / `table`.
    Every output row gets at least one (possibly empty) item, so the kernel also writes the zeros.
    group_slots: 0 = split rows go through the partial buffer + finalize launch; G > 0 = split rows
    are combined inside one workgroup (a row is cut into at most G balanced pieces; one workgroup
    runs one block, see `group_slots_for`).
    """
    dev = out_row.device
    E = int(out_row.numel())
    if chunk is None:
        chunk = auto_chunk(E)
    if E >= 2 ** 31 - 1 or n_out >= 2 ** 31 - 1 or n_table >= 2 ** 31 - 1:
        raise ValueError('graph too large for int32 plans')
    if E:
        lo, hi = int(out_row.min()), int(out_row.max())
        tlo, thi = int(table_row.min()), int(table_row.max())
        if lo < 0 or hi >= n_out or tlo < 0 or thi >= n_table:
            raise IndexError('edge index out of range: out rows [%d,%d] of %d, table rows [%d,%d] of %d'
                             % (lo, hi, n_out, tlo, thi, n_table))
    order = torch.sort(out_row, stable=True).indices
    counts = torch.bincount(out_row, minlength=n_out) if E else torch.zeros(n_out, dtype=torch.long, device=dev)
    row_ptr = torch.zeros(n_out + 1, dtype=torch.long, device=dev)
    row_ptr[1:] = torch.cumsum(counts, 0)

    n_chunks = torch.clamp((counts + chunk - 1) // chunk, min=1)            # items per row
    piece_len = torch.full_like(n_chunks, chunk)
    if group_slots:
        n_chunks = torch.clamp(n_chunks, max=group_slots)                    # hub rows: longer pieces, one workgroup
        piece_len = torch.clamp((counts + n_chunks - 1) // n_chunks, min=1)  # balanced pieces
    item_ptr = torch.zeros(n_out + 1, dtype=torch.long, device=dev)
    item_ptr[1:] = torch.cumsum(n_chunks, 0)
    n_items = int(item_ptr[-1])
    item_row = torch.repeat_interleave(torch.arange(n_out, device=dev), n_chunks)
    local = torch.arange(n_items, device=dev) - item_ptr[item_row]
    row_end = row_ptr[item_row + 1]
    begin = torch.minimum(row_ptr[item_row] + local * piece_len[item_row], row_end)
    end = torch.minimum(begin + piece_len[item_row], row_end)
    direct = n_chunks[item_row] == 1

    row_id = table_row[order].to(torch.int32).contiguous()
    w = None if edge_w is None else edge_w[order].to(torch.float32).contiguous()
    split = torch.nonzero(n_chunks > 1).view(-1)

    if group_slots:
        G = int(group_slots)
        pieces = n_chunks[split].tolist()
        block, offset, n_blocks = pack_blocks(pieces, G)
        grouped = torch.zeros((n_blocks * G, 4), dtype=torch.long, device=dev)
        grouped[:, 3] = ITEM_NULL
        if split.numel():
            base = torch.tensor(block, device=dev) * G + torch.tensor(offset, device=dev)      # first slot per row
            pc = n_chunks[split]
            src = torch.repeat_interleave(item_ptr[split], pc) + \
                (torch.arange(int(pc.sum()), device=dev) - torch.repeat_interleave(torch.cumsum(pc, 0) - pc, pc))
            dst = torch.repeat_interleave(base, pc) + (src - torch.repeat_interleave(item_ptr[split], pc))
            lead = src == torch.repeat_interleave(item_ptr[split], pc)
            flags = torch.where(lead, ITEM_PIECE | ITEM_LEADER | (torch.repeat_interleave(pc, pc) << 8),
                                torch.full_like(src, ITEM_PIECE))
            grouped[dst] = torch.stack([begin[src], end[src], item_row[src], flags], dim=1)
        d_idx = torch.nonzero(direct).view(-1)
        by_len = torch.sort((end - begin)[d_idx], descending=True, stable=True).indices
        d_idx = d_idx[by_len]
        plain = torch.stack([begin[d_idx], end[d_idx], item_row[d_idx], torch.full_like(d_idx, ITEM_DIRECT)], dim=1)
        items = torch.cat([grouped, plain], 0).to(torch.int32).contiguous()
        split_rows = torch.zeros((0, 3), dtype=torch.int32, device=dev)
        return GatherPlan(n_out, n_table, row_id, w, items, split_rows, 0, order, chunk, tag, G)

    slot = torch.cumsum((~direct).long(), 0) - 1                             # slot id of split items
    target = torch.where(direct, item_row, slot)
    n_slots = int((~direct).sum())

    by_len = torch.sort(end - begin, descending=True, stable=True).indices
    items = torch.stack([begin, end, target, direct.long()], dim=1)[by_len].to(torch.int32).contiguous()

    if split.numel():
        first = slot[item_ptr[split]]
        split_rows = torch.stack([split, first, first + n_chunks[split]], dim=1).to(torch.int32).contiguous()
    else:
        split_rows = torch.zeros((0, 3), dtype=torch.int32, device=dev)
    return GatherPlan(n_out, n_table, row_id, w, items, split_rows, n_slots, order, chunk, tag)


def build_gather_plan_segmented(out_row, table_row, segment, n_out, n_table, chunk=DEFAULT_CHUNK, tag=''):
    """`build_gather_plan` for tables far larger than the caches (a D-D forward pass over
    Y = [R N, d] on a big graph: 10 GB in BASELINE config 5), where the ORDER in which the work items
    run decides the HBM traffic.

    segment: int64 [E], a coarse id of the table region an edge reads (relation block =
    relation // B with B relations x N rows x d floats ~ a quarter of the 256 MB Infinity Cache),
    non-decreasing along the caller's edge order inside every output row (TIP's edge lists are
    grouped by relation).  Every (output row, segment) group is cut into items of <= chunk edges and
    the items are launched SEGMENT BY SEGMENT (inside a segment by decreasing length): all items in
    flight at any time read the same few relations' rows, so a row of Y that several edges gather
    (2.5 on average in config 5) comes from HBM once and from the Infinity Cache afterwards.  The
    default plan launches row by row: every item then spans ~50 relations of one output row and the
    launch touches the whole table at random -- each row crosses the fabric once PER EDGE.
    Rows consist of several items (one per segment at least): their pieces are added in slot order by
    `tipk_gather_sum_finalize` (deterministic), exactly as for split rows of the default plan."""
    dev = out_row.device
    E = int(out_row.numel())
    if chunk is None:
        chunk = auto_chunk(E)
    if E >= 2 ** 31 - 1 or n_out >= 2 ** 31 - 1 or n_table >= 2 ** 31 - 1:
        raise ValueError('graph too large for int32 plans')
    if E:
        lo, hi = int(out_row.min()), int(out_row.max())
        tlo, thi = int(table_row.min()), int(table_row.max())
        if lo < 0 or hi >= n_out or tlo < 0 or thi >= n_table:
            raise IndexError('edge index out of range: out rows [%d,%d] of %d, table rows [%d,%d] of %d'
                             % (lo, hi, n_out, tlo, thi, n_table))
    n_seg = (int(segment.max()) + 1) if E else 1
    order = torch.sort(out_row, stable=True).indices
    key = out_row[order] * n_seg + segment[order]
    if E and bool((key[1:] < key[:-1]).any()):
        raise ValueError('segment ids must be non-decreasing inside every output row (edges grouped by relation)')
    gkey, gcount = torch.unique_consecutive(key, return_counts=True)
    grow = gkey // n_seg
    # rows without edges still get one (empty, direct) item so that the kernel writes their zeros
    has = torch.zeros(n_out, dtype=torch.bool, device=dev)
    has[grow] = True
    empty = torch.nonzero(~has).view(-1)
    gbegin = torch.cumsum(gcount, 0) - gcount
    if empty.numel():
        gkey = torch.cat([gkey, empty * n_seg])
        gcount = torch.cat([gcount, torch.zeros_like(empty)])
        gbegin = torch.cat([gbegin, torch.zeros_like(empty)])
        o = torch.sort(gkey, stable=True).indices
        gkey, gcount, gbegin = gkey[o], gcount[o], gbegin[o]
        grow = gkey // n_seg
    gseg = gkey % n_seg
    n_chunks = torch.clamp((gcount + chunk - 1) // chunk, min=1)
    item_grp = torch.repeat_interleave(torch.arange(gkey.numel(), device=dev), n_chunks)
    first_item = torch.cumsum(n_chunks, 0) - n_chunks
    local = torch.arange(item_grp.numel(), device=dev) - first_item[item_grp]
    gend = gbegin + gcount
    begin = torch.minimum(gbegin[item_grp] + local * chunk, gend[item_grp])
    end = torch.minimum(begin + chunk, gend[item_grp])
    item_row = grow[item_grp]
    items_per_row = torch.bincount(item_row, minlength=n_out)
    direct = items_per_row[item_row] == 1
    slot = torch.cumsum((~direct).long(), 0) - 1                             # (row, piece) order: consecutive per row
    target = torch.where(direct, item_row, slot)
    n_slots = int((~direct).sum())
    # launch order: segment-major, longest first inside a segment
    okey = gseg[item_grp] * (chunk + 1) + (chunk - (end - begin))
    by = torch.sort(okey, stable=True).indices
    items = torch.stack([begin, end, target, direct.long()], dim=1)[by].to(torch.int32).contiguous()
    row_id = table_row[order].to(torch.int32).contiguous()
    split = torch.nonzero(items_per_row > 1).view(-1)
    if split.numel():
        row_first_item = torch.cumsum(items_per_row, 0) - items_per_row
        first = slot[row_first_item[split]]
        split_rows = torch.stack([split, first, first + items_per_row[split]], dim=1).to(torch.int32).contiguous()
    else:
        split_rows = torch.zeros((0, 3), dtype=torch.int32, device=dev)
    plan = GatherPlan(n_out, n_table, row_id, None, items, split_rows, n_slots, order, chunk, tag)
    # item range of every segment in launch order (host list): a caller may run the plan segment by segment, handing
    # each launch only the table rows of that segment (tip_amd/ops.py `_RGCN.forward`: the rows of Y are produced
    # block by block and consumed out of the Infinity Cache)
    per_seg = torch.bincount(gseg[item_grp], minlength=n_seg)
    plan.seg_item_ptr = [0] + torch.cumsum(per_seg, 0).tolist()
    return plan


class CsrPlan(object):
    """out[r] = sum of table rows over row r's edge range (include/tipk.h section 1c): edges stably
    sorted by output row; row_ptr int32 [n_out + 1], row_id int32 [E] (table row per edge)."""

    def __init__(self, n_out, n_table, row_ptr, row_id, tag=''):
        self.n_out, self.n_table, self.row_ptr, self.row_id, self.tag = int(n_out), int(n_table), row_ptr, row_id, tag
        self.n_edges = int(row_id.numel())


def build_csr_plan(out_row, table_row, n_out, n_table, tag=''):
    E = int(out_row.numel())
    if E >= 2 ** 31 - 1 or n_out >= 2 ** 31 - 2 or n_table >= 2 ** 31 - 1:
        raise ValueError('graph too large for int32 plans')
    if E:
        lo, hi = int(out_row.min()), int(out_row.max())
        tlo, thi = int(table_row.min()), int(table_row.max())
        if lo < 0 or hi >= n_out or tlo < 0 or thi >= n_table:
            raise IndexError('edge index out of range')
    order = torch.sort(out_row, stable=True).indices
    counts = torch.bincount(out_row, minlength=n_out) if E else torch.zeros(n_out, dtype=torch.long, device=out_row.device)
    row_ptr = torch.zeros(n_out + 1, dtype=torch.int32, device=out_row.device)
    row_ptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
    return CsrPlan(n_out, n_table, row_ptr, table_row[order].to(torch.int32).contiguous(), tag)


def execute_csr_reference(plan, table):
    """Pure-torch interpretation of a CsrPlan (CPU unit tests only)."""
    ptr = plan.row_ptr.long()
    row_of = torch.repeat_interleave(torch.arange(plan.n_out), ptr[1:] - ptr[:-1])
    out = torch.zeros((plan.n_out, table.shape[1]), dtype=table.dtype)
    return out.index_add_(0, row_of, table[plan.row_id.long()])


def relations_per_segment(n_nodes, d, budget_bytes=64 << 20):
    """Relations per segment of `build_gather_plan_segmented`: their rows of Y (N x d fp32 each)
    take about `budget_bytes` -- a quarter of the Infinity Cache, so that the two or three segments in
    flight at a time stay resident next to the streamed ids and partial sums."""
    return max(1, int(budget_bytes // max(1, n_nodes * d * 4)))


def execute_plan_reference(plan, table, row_scale=None):
    """Pure-torch interpretation of a plan (item by item semantics, vectorised).  Used by the CPU
    unit tests of the plan builder and of the host logic; NOT used by the product path."""
    d = table.shape[1]
    it = plan.items.long()
    lens = it[:, 1] - it[:, 0]
    item_of_edge = torch.repeat_interleave(torch.arange(it.shape[0], device=table.device), lens)
    starts = torch.cumsum(lens, 0) - lens
    edge_pos = it[item_of_edge, 0] + (torch.arange(int(lens.sum()), device=table.device) - starts[item_of_edge])
    rows = table[plan.row_id.long()[edge_pos]]
    if plan.edge_w is not None:
        rows = rows * plan.edge_w[edge_pos].unsqueeze(1)
    per_item = torch.zeros((it.shape[0], d), dtype=table.dtype, device=table.device).index_add_(0, item_of_edge, rows)
    out = torch.zeros((plan.n_out, d), dtype=table.dtype, device=table.device)
    direct = (it[:, 3] & ITEM_DIRECT) != 0
    out[it[direct, 2]] = per_item[direct]
    if plan.n_slots:
        part = it[:, 3] == 0
        partial = torch.zeros((plan.n_slots, d), dtype=table.dtype, device=table.device)
        partial[it[part, 2]] = per_item[part]
        for r, a, b in plan.split_rows.tolist():
            out[r] = partial[a:b].sum(0)
    if plan.group_slots:
        G = plan.group_slots
        for i in torch.nonzero((it[:, 3] & ITEM_LEADER) != 0).view(-1).tolist():
            cnt = int(it[i, 3]) >> 8
            assert i // G == (i + cnt - 1) // G, 'pieces of a row must stay inside one block'
            assert bool(((it[i:i + cnt, 3] & ITEM_PIECE) != 0).all()) and bool((it[i:i + cnt, 2] == it[i, 2]).all())
            acc = per_item[i].clone()
            for j in range(1, cnt):                  # the kernel's order: leader first, then slot by slot
                acc = acc + per_item[i + j]
            out[it[i, 2]] = acc
    if row_scale is not None:
        out = out * row_scale.unsqueeze(1)
    return out


# ---------------------------------------------------------------------------------------------
# relation-local plans (include/tipk.h section 1b): LDS-resident D-D aggregation
# ---------------------------------------------------------------------------------------------
class RelPlan(object):
    """Device arrays of `tipk_rel_gather` for one direction of a multi-relational graph (layout:
    include/tipk.h section 1b).  Per work unit: node_at, runs (+ unit_rel, unit_npos, rel_idx_off,
    rel_len, which the kernel reads through `unit_meta`, the descriptors in workgroup order);
    wg_rel_ptr = range of every workgroup in unit_meta, wg_rels = the same order as unit ids."""

    def __init__(self, n_nodes, n_rel, n_wg, node_at, rel_idx_off, rel_len, idx, runs, wg_rel_ptr, wg_rels,
                 unit_rel, unit_npos, unit_meta=None, idx_unit=1):
        self.n_nodes, self.n_rel, self.n_wg = int(n_nodes), int(n_rel), int(n_wg)
        self.idx_unit = int(idx_unit)             # idx holds node * idx_unit (pre-scaled LDS row offsets)
        self.node_at, self.rel_idx_off, self.rel_len, self.idx, self.runs = node_at, rel_idx_off, rel_len, idx, runs
        self.wg_rel_ptr, self.wg_rels = wg_rel_ptr, wg_rels
        self.unit_rel, self.unit_npos = unit_rel, unit_npos
        self.n_units = int(unit_rel.numel())
        if unit_meta is None:
            u = wg_rels.long()
            off = rel_idx_off[u]
            unit_meta = torch.stack([u, unit_rel[u].long(), unit_npos[u].long(), rel_len[u].long(),
                                     off & 0xffffffff, off >> 32, torch.zeros_like(u), torch.zeros_like(u)], dim=1)
            unit_meta = (unit_meta & 0xffffffff).to(torch.int64)
            unit_meta = torch.where(unit_meta >= 2 ** 31, unit_meta - 2 ** 32, unit_meta).to(torch.int32).contiguous()
        self.unit_meta = unit_meta

    def to(self, device):
        return RelPlan(self.n_nodes, self.n_rel, self.n_wg, *[t.to(device) for t in (
            self.node_at, self.rel_idx_off, self.rel_len, self.idx, self.runs, self.wg_rel_ptr, self.wg_rels,
            self.unit_rel, self.unit_npos, self.unit_meta)], idx_unit=self.idx_unit)


def assign_relations(sizes, n_wg, fixed_cost=0):
    """Greedy longest-processing-time deal of relations to workgroups (cost = edges + fixed_cost).
    -> (wg_rel_ptr [n_wg+1], wg_rels [R]) int32 CPU tensors; deterministic."""
    sizes = [int(s) for s in sizes]
    order = sorted(range(len(sizes)), key=lambda r: (-sizes[r], r))
    import heapq
    heap = [(0, w) for w in range(n_wg)]
    heapq.heapify(heap)
    parts = [[] for _ in range(n_wg)]
    for r in order:
        load, w = heapq.heappop(heap)
        parts[w].append(r)
        heapq.heappush(heap, (load + sizes[r] + fixed_cost, w))
    ptr = [0]
    flat = []
    for p in parts:
        flat.extend(p)
        ptr.append(len(flat))
    return torch.tensor(ptr, dtype=torch.int32), torch.tensor(flat, dtype=torch.int32)


# ds_read_b128 serves a wave64 in four groups of 16 lanes (MI355X_MICROARCH.md, LDS): only lanes of one
# group can conflict.  Lanes of the first half-wave: group 0 = {0-3, 12-15, 20-27}, group 1 = the rest.
_B128_GROUP_OF_LANE = [0 if (l % 32) in (0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27) else 1 for l in range(64)]


def bank_rotation(lanes):
    """(C, rot[k]) for slots of `lanes` lanes (one float4 each: unpadded rows of lanes * 16 bytes).

    A row's position inside the 256-byte bank row is its bank CLASS = node mod C, C = 256 / row bytes;
    a 16-lane group of one ds_read_b128 is conflict-free iff the rows (or half rows) it reads fall into
    different classes.  rot[k] is the offset slot k (index inside its wavefront) gives its class
    sequence: the plan sorts the ids of the run that slot k walks by (class - rot[k]) mod C, so the slots
    of a group that advance in lock step read classes that differ by their rot.  lanes = 8: a 128-byte row
    spans two groups (64 bytes each); the slot pairs (0,3) and (1,2) of every half-wave share their groups
    with opposite halves, so they must differ in PARITY.  lanes >= 16: a row covers a whole bank row --
    never a conflict between slots (C = 1)."""
    per_wave = 64 // lanes
    if lanes >= 16:
        return 1, [0] * per_wave
    if lanes == 8:
        return 2, [(0, 0, 1, 1)[k % 4] for k in range(per_wave)]
    rot, seen = [], {}
    for k in range(per_wave):
        g = (k * lanes) // 32 * 2 + _B128_GROUP_OF_LANE[k * lanes]           # (half-wave, group)
        rot.append(seen.get(g, 0))
        seen[g] = rot[-1] + 1
    return 16 // lanes, rot


def build_rel_plan(out_node, tab_node, rel, n_nodes, n_rel, n_wg=256, fixed_cost=2048, backward=False,
                   max_unit=None, lanes=None, unit_cap=None):
    """Relation-local plan for  result[r, o] = sum_{e in r: out_node[e]=o} table_r[tab_node[e]].

    out_node / tab_node / rel: int64 [E]; nodes < 65536.  Inside every relation the output nodes
    are ordered by decreasing run length and the edges are sorted by the position of their output
    node, so the edges of one (relation, node) pair form one contiguous run and neighbouring
    positions have runs of similar length.

    Work units: a relation with more than `max_unit` (padded) ids is dealt position by position to
    k = ceil(ids / max_unit) units, so that no single relation sets the length of the launch
    (default max_unit: 1.2x / 0.55x the per-workgroup average, forward / backward).  backward: the plan drives the
    transposed pass, where every (relation, node) row must be written -> units also walk their
    empty positions; forward units stop at their last non-empty position.

    unit_cap: upper bound for the automatic max_unit (the kernel's id chunk, `tipk_rel_gather_chunk`).
    lanes: lanes per slot of the launch this plan is for (= columns of one column block / 4).  Then
    (i) the ids INSIDE every run are ordered for conflict-free LDS reads (`bank_rotation`: position p
    of a unit is walked by slot (p mod S, snaked) of its band, S = 1024 / lanes), and (ii) idx holds
    pre-scaled row offsets node * idx_unit.  None: plain ids in edge order (any launch shape)."""
    dev = out_node.device
    assert n_nodes <= 65535
    E = int(out_node.numel())
    N = n_nodes
    cnt_nodes = torch.bincount(rel * N + out_node, minlength=n_rel * N).view(n_rel, N)
    order_r = torch.sort(cnt_nodes, dim=1, descending=True, stable=True).indices          # [R, N] position -> node
    pos_of = torch.empty_like(order_r)
    pos_of.scatter_(1, order_r, torch.arange(N, device=dev).expand(n_rel, N).contiguous())
    cnt_r = torch.gather(cnt_nodes, 1, order_r)                                           # run length per position
    size_r = ((cnt_r + 7) // 8 * 8).sum(1)                                                # padded ids per relation
    if max_unit is None:
        # measured on BioSNAP (tools/bench_relgather.py): forward units re-stage the relation's table, so
        # only relations above ~1.2x the per-workgroup average are cut; backward units are cheap (0.55x)
        mean_load = int(size_r.sum()) // max(n_wg, 1)
        max_unit = max(4096, int(mean_load * (0.55 if backward else 1.2)))
        if unit_cap:                               # ids the kernel stages per pass: a larger unit reloads synchronously
            max_unit = min(max_unit, int(unit_cap))
    k_r = torch.clamp((size_r + max_unit - 1) // max_unit, min=1, max=max(N, 1))          # units per relation
    unit_base = torch.cumsum(k_r, 0) - k_r
    U = int(k_r.sum())
    unit_rel = torch.repeat_interleave(torch.arange(n_rel, device=dev), k_r)              # [U]
    unit_j = torch.arange(U, device=dev) - unit_base[unit_rel]                            # share index inside the relation
    unit_k = k_r[unit_rel]
    # position q of unit (r, j) is position q * k + j of relation r
    P = torch.arange(N, device=dev).unsqueeze(0) * unit_k.unsqueeze(1) + unit_j.unsqueeze(1)      # [U, N]
    valid = P < N
    Pc = torch.where(valid, P, torch.zeros_like(P))
    node_at = torch.where(valid, order_r[unit_rel.unsqueeze(1), Pc], torch.zeros_like(P))
    cnt = torch.where(valid, cnt_r[unit_rel.unsqueeze(1), Pc], torch.zeros_like(P))       # [U, N] run lengths
    npos = valid.sum(1) if backward else (cnt > 0).sum(1)
    # every run is padded to a multiple of 8 ids with the sentinel id n_nodes (its table row is
    # zero in the kernel), so runs and unit segments start 16-byte aligned and need no masks
    cnt8 = (cnt + 7) // 8 * 8
    begin8 = torch.cumsum(cnt8, 1) - cnt8                                                 # relative to the unit
    runs = torch.stack([begin8, cnt8], dim=2).to(torch.int32).contiguous()
    unit_sizes = cnt8.sum(1)
    off = torch.cumsum(unit_sizes, 0) - unit_sizes
    total = int(unit_sizes.sum()) + 8 + 2048 * 8     # slack: the kernel prefetches whole 16 K id chunks unconditionally
    # destination of every edge: unit offset + run begin + rank inside the run
    p_e = pos_of.view(-1)[rel * N + out_node]
    u_e = unit_base[rel] + p_e % k_r[rel]
    q_e = p_e // k_r[rel]                                                                 # position inside the unit
    key = u_e * N + q_e
    idx_unit = 1
    if lanes:
        n_slots = 1024 // lanes
        band, j = q_e // n_slots, q_e % n_slots
        slot = torch.where(band % 2 == 1, n_slots - 1 - j, j)                             # the kernel's snake deal
        n_cls, rot = bank_rotation(lanes)
        rot_t = torch.tensor(rot, device=dev)[slot % (64 // lanes)]
        cls = (tab_node % n_cls - rot_t) % n_cls
        order = torch.sort(key * n_cls + cls, stable=True).indices
        row_bytes = lanes * 16
        while idx_unit * 2 <= row_bytes and N * idx_unit * 2 <= 65535:
            idx_unit *= 2
    else:
        order = torch.sort(key, stable=True).indices
    skey = key[order]
    run_first = torch.cumsum(cnt.view(-1), 0) - cnt.view(-1)                              # first sorted edge of a run
    rank = torch.arange(E, device=dev) - run_first[skey]
    dest = off[skey // N] + begin8.view(-1)[skey] + rank
    idx32 = torch.full((total,), N * idx_unit, dtype=torch.int32, device=dev)
    idx32[dest] = (tab_node[order] * idx_unit).to(torch.int32)
    idx = idx32.to(torch.uint16).contiguous()
    if E and int(unit_sizes.max()) >= 2 ** 19:
        raise ValueError('a work unit has %d ids: the kernel packs run offsets / 8 into 16 bits (max 2^19 - 8 ids per unit); '
                         'lower max_unit' % int(unit_sizes.max()))
    wg_ptr, wg_rels = assign_relations(unit_sizes.tolist(), n_wg, fixed_cost)
    return RelPlan(N, n_rel, n_wg, node_at.to(torch.int32).to(torch.uint16).contiguous(),
                   off.to(torch.int64).contiguous(), unit_sizes.to(torch.int32).contiguous(), idx, runs,
                   wg_ptr.to(dev), wg_rels.to(dev), unit_rel.to(torch.int32).contiguous(),
                   npos.to(torch.int32).contiguous(), idx_unit=idx_unit)


class StreamPlan(object):
    """Device arrays of `tipk_stream_gather` (layout: include/tipk.h section 1d): out rows <- sums of table rows."""

    def __init__(self, n_rows, n_table, n_wg, lanes, piece, wave_ptr, cells, ids, zero_ptr, zero_rows, idx_unit,
                 row_used=None, n_nodes=None, n_rel=None):
        self.n_rows, self.n_table, self.n_wg, self.lanes, self.piece = int(n_rows), int(n_table), int(n_wg), int(lanes), int(piece)
        self.wave_ptr, self.cells, self.ids, self.zero_ptr, self.zero_rows = wave_ptr, cells, ids, zero_ptr, zero_rows
        self.idx_unit = int(idx_unit)
        self.n_bands = int(cells.shape[0])
        # the (relation, node) form of `build_stream_plan`: rows = relation * n_nodes + node
        self.row_used = row_used                  # int32 [ceil(R / 32), N] bit mask of the rows with edges (tipk.h section 2b)
        self.symmetric = False                    # pair-form plans: built from the edges with source <= destination only
        self.links = None                         # pair-form plans: link words [sources padded to 8, ceil(N / 32)] (layers.pair_link_words)
        self.n_edges = 0                          # edges the plan walks
        self.n_nodes, self.n_rel = n_nodes, n_rel
        self.compact = None                       # CompactRows: the rows are the node-major compact numbering (tipk.h section 2d)
        self.row_bytes = self.lanes * 16          # bytes of a table row of one column block in LDS
        self.dyc = {}                             # compact plans: (width, device) -> the dY buffer [n_rows + 1, width] (ops.rel_stream_bwd)

    def to(self, device):
        mv = lambda t: None if t is None else t.to(device)
        sp = StreamPlan(self.n_rows, self.n_table, self.n_wg, self.lanes, self.piece, mv(self.wave_ptr), mv(self.cells),
                        mv(self.ids), mv(self.zero_ptr), mv(self.zero_rows), self.idx_unit, mv(self.row_used),
                        self.n_nodes, self.n_rel)
        sp.symmetric, sp.n_edges, sp.row_bytes = self.symmetric, self.n_edges, self.row_bytes
        sp.links = mv(self.links)
        sp.compact = None if self.compact is None else self.compact.to(device)
        return sp


STREAM_BAND_OVERHEAD = 2.0     # what a band costs besides its steps (record fetch, row store), in steps


STREAM_WIDE_STEPS = 16        # a run with more steps than this is cut into 2, 4 or 8 sub-runs walked side by side


def build_stream_plan_rows(out_row, tab_row, n_rows, n_table, n_wg, lanes, piece=4, wide_steps=None, row_bytes=None):
    """Wave-stream plan for  out[o] = sum_{e: out_row[e] = o} table[tab_row[e]],  o < n_rows, table rows < n_table.

    The runs (one per output row with edges) are sorted by decreasing length and taken SPW = 64 / lanes at
    a time: such a GROUP is what the slots of one wavefront walk side by side, so its runs should be
    equally long (no lanes idling behind a hub run) -- after the sort they are.  A group whose longest run
    has s steps of 8 ids occupies ceil(s / piece) consecutive BANDS of its wavefront (a run continues in
    the same slot, its sum stays in registers).  Groups are dealt to the n_wg * 16 wavefronts by the
    longest-processing-time rule on  steps + STREAM_BAND_OVERHEAD * bands; rows without edges are dealt
    evenly.  Inside a run the ids are ordered for conflict-free LDS reads (`bank_rotation`) and pre-scaled
    (idx_unit) exactly as in `build_rel_plan`.

    WIDE runs: a slot walks its run as a chain of dependent steps, so the longest run of the graph (BioSNAP:
    a drug pair linked by 475 relations = 60 steps) set the length of the launch once everything else was
    balanced (measured: capping runs at 24 steps, 4 % of the work, took the pair gather from 21.5 to 15.2 us).
    A run with more than `wide_steps` steps is therefore cut into k = 2, 4 or 8 sub-runs that sit in k
    adjacent, k-aligned slots of ONE group; their partial sums are added in a fixed tree order by the kernel
    (cell bits 30-31 = log2 k on the set's last band) and the first slot writes the row.

    row_bytes: bytes of a table row in LDS (default lanes * 16: a float4 per lane); 8 = the 2-column blocks of a table
    too tall for 16-byte rows (the P-P graph: lanes = 1, a float2 per lane)."""
    import heapq
    dev = out_row.device
    T, S, W = int(n_table), 64 // int(lanes), int(n_wg) * 16
    assert T <= 65535 and n_rows < 2 ** 24, 'cells hold the output row in 24 bits, ids are 16-bit'
    if wide_steps is None:
        wide_steps = STREAM_WIDE_STEPS
    E = int(out_row.numel())
    cnt_rows = torch.bincount(out_row, minlength=n_rows)
    zero_rows = torch.nonzero(cnt_rows == 0).flatten()
    run_row = torch.nonzero(cnt_rows > 0).flatten()                        # runs in row order
    run_cnt = cnt_rows[run_row]
    n_runs = int(run_row.numel())
    run_steps = (run_cnt + 7) // 8
    # sub-runs per run (1 = plain) and steps per sub-run
    kmax = min(S, 8)
    need = (run_steps + wide_steps - 1) // wide_steps
    k_run = torch.ones_like(run_steps)
    for kk in (2, 4, 8):
        if kk <= kmax:
            k_run = torch.where(need > kk // 2, torch.full_like(k_run, kk), k_run)
    q_run = (run_steps + k_run - 1) // k_run                               # steps of the (largest) sub-runs
    # ---- the slot list ("virtual runs") in layout order: wide sets first (largest k first, k-aligned inside groups
    # of S slots), then the plain runs by decreasing steps (rows ascending inside a class: the rows a wavefront
    # writes together are neighbours in the output more often than not)
    v_run, v_pos = [], []                                                   # per virtual run: original run (-1 = idle slot), position in its set
    for kk in (8, 4, 2):
        sel = torch.nonzero(k_run == kk).flatten()
        if sel.numel() == 0:
            continue
        sel = sel[torch.sort(q_run[sel], descending=True, stable=True).indices]
        v_run.append(torch.repeat_interleave(sel, kk))
        v_pos.append(torch.arange(kk, device=dev).repeat(sel.numel()))
        pad = (-int(sel.numel()) * kk) % S
        v_run.append(torch.full((pad,), -1, dtype=torch.int64, device=dev))
        v_pos.append(torch.zeros(pad, dtype=torch.int64, device=dev))
    plain = torch.nonzero(k_run == 1).flatten()
    plain = plain[torch.sort(run_steps[plain], descending=True, stable=True).indices]
    v_run.append(plain)
    v_pos.append(torch.zeros(plain.numel(), dtype=torch.int64, device=dev))
    pad = (-sum(int(t.numel()) for t in v_run)) % S
    v_run.append(torch.full((pad,), -1, dtype=torch.int64, device=dev))
    v_pos.append(torch.zeros(pad, dtype=torch.int64, device=dev))
    v_run, v_pos = torch.cat(v_run), torch.cat(v_pos)
    V = int(v_run.numel())
    G = V // S
    real = v_run >= 0
    vr = torch.clamp(v_run, min=0)
    v_k = torch.where(real, k_run[vr], torch.ones_like(vr)) if n_runs else torch.ones_like(vr)
    v_q = torch.where(real, q_run[vr], torch.zeros_like(vr)) if n_runs else torch.zeros_like(vr)
    v_tot = torch.where(real, run_steps[vr], torch.zeros_like(vr)) if n_runs else torch.zeros_like(vr)
    v_steps = torch.clamp(v_tot - v_pos * v_q, min=0)
    v_steps = torch.minimum(v_steps, v_q)                                   # steps of this sub-run
    v_nb = torch.where(real, torch.clamp((v_q + piece - 1) // piece, min=1), torch.zeros_like(v_q))   # bands of its set
    v_klog = torch.where(v_k >= 8, 3, torch.where(v_k >= 4, 2, torch.where(v_k >= 2, 1, 0)))
    v_rowid = torch.where(real, run_row[vr], torch.zeros_like(vr)) if n_runs else torch.zeros_like(vr)
    # ---- groups -> wavefronts
    g_steps = v_q.view(G, S).max(1).values if G else v_q.new_zeros(0)
    g_bands = torch.clamp((g_steps + piece - 1) // piece, min=1)
    cost_t = g_steps.double() + STREAM_BAND_OVERHEAD * g_bands.double()
    by_cost = torch.sort(cost_t, descending=True, stable=True).indices.cpu().tolist()
    cost = cost_t.cpu().tolist()
    # longest-processing-time deal
    # (giving every wavefront a contiguous range of output rows instead -- length-sorted only inside windows of
    # 645 ... 32 768 rows, so that its writes sweep the output front to back -- measured no faster inside the step)
    heap = [(0.0, w) for w in range(W)]
    wave_of = [0] * G
    for g in by_cost:
        load, w = heapq.heappop(heap)
        wave_of[g] = w
        heapq.heappush(heap, (load + cost[g], w))
    wave_of = torch.tensor(wave_of, dtype=torch.int64, device=dev)
    g_order = torch.sort(wave_of, stable=True).indices                   # layout order: wavefront by wavefront
    bands_l = g_bands[g_order]
    band0_l = torch.cumsum(bands_l, 0) - bands_l
    n_bands = int(bands_l.sum()) if G else 0
    g_band0 = torch.empty(G, dtype=torch.int64, device=dev)
    g_band0[g_order] = band0_l
    per_wave = torch.zeros(W, dtype=torch.int64, device=dev).index_add_(0, wave_of, g_bands)
    wave_ptr = torch.cat([per_wave.new_zeros(1), torch.cumsum(per_wave, 0)])
    # ---- cells
    grp = torch.repeat_interleave(g_order, bands_l)                       # group of every band
    k = (torch.arange(n_bands, device=dev) - torch.repeat_interleave(band0_l, bands_l)).unsqueeze(1)
    pick = lambda t: t.view(G, S)[grp]                                    # [n_bands, S]
    st, nb_set, pos, klog, rid, rl = pick(v_steps), pick(v_nb), pick(v_pos), pick(v_klog), pick(v_rowid), pick(real)
    length = torch.clamp(st - k * piece, min=0, max=piece)
    final = k == nb_set - 1
    first = rl & (k == 0)
    last = rl & final & (pos == 0)
    active = rl & (k < nb_set)
    cells = torch.where(active, rid | (length << 24) | (first.long() << 28) | (last.long() << 29)
                        | (torch.where(final, klog, torch.zeros_like(klog)) << 30), torch.zeros_like(st))
    # ---- ids
    idx_unit = 1
    row_bytes = lanes * 16 if row_bytes is None else int(row_bytes)
    assert row_bytes == lanes * 16 or (row_bytes == 8 and lanes == 1)
    while idx_unit * 2 <= row_bytes and T * idx_unit * 2 <= 65535:
        idx_unit *= 2
    v_first = torch.full((max(n_runs, 1),), -1, dtype=torch.int64, device=dev)      # virtual index of a run's first sub-run
    head = real & (v_pos == 0)
    v_first[v_run[head]] = torch.nonzero(head).flatten()
    run_of_row = torch.full((n_rows,), -1, dtype=torch.int64, device=dev)
    run_of_row[run_row] = torch.arange(n_runs, device=dev)
    ri = run_of_row[out_row]                                              # run of every edge
    o1 = torch.sort(ri, stable=True).indices
    ri1 = ri[o1]
    run_first = torch.cumsum(run_cnt, 0) - run_cnt
    j1 = torch.arange(E, device=dev) - run_first[ri1]                     # rank inside the run (edge order)
    q8 = q_run[ri1] * 8
    ve = v_first[ri1] + j1 // q8                                          # virtual run of every edge
    if row_bytes == 8:
        # 8-byte rows: a row sits in bank pair (id mod 32); the 64 slots of a wavefront advance in lock step, slot k
        # starts its class sequence k mod 32 later -- any 32 neighbouring lanes then read 32 different bank pairs
        n_cls, rot = 32, [k % 32 for k in range(S)]
    else:
        n_cls, rot = bank_rotation(lanes)
    tab1 = tab_row[o1]
    cls = (tab1 % n_cls - torch.tensor(rot, device=dev)[ve % S]) % n_cls
    o2 = torch.sort(ve * n_cls + cls, stable=True).indices
    ve2 = ve[o2]
    v_cnt = torch.bincount(ve2, minlength=V) if E else torch.zeros(V, dtype=torch.int64, device=dev)
    v_start = torch.cumsum(v_cnt, 0) - v_cnt
    jr = torch.arange(E, device=dev) - v_start[ve2]                       # rank inside the sub-run (bank order)
    step = jr // 8
    band = g_band0[ve2 // S] + step // piece
    dest = ((band * piece + step % piece) * S + ve2 % S) * 8 + jr % 8
    ids32 = torch.full((max(n_bands, 1) * piece * S * 8,), T * idx_unit, dtype=torch.int32, device=dev)
    ids32[dest] = (tab1[o2] * idx_unit).to(torch.int32)
    nz = int(zero_rows.numel())
    zero_ptr = (torch.arange(W + 1, device=dev) * nz) // W
    cells_u = torch.where(cells >= 2 ** 31, cells - 2 ** 32, cells).to(torch.int32)
    if zero_rows.numel() == 0:
        zero_rows = torch.zeros(1, dtype=torch.int64, device=dev)
    sp = StreamPlan(n_rows, T, n_wg, lanes, piece, wave_ptr.to(torch.int32).contiguous(),
                    cells_u.view(n_bands, S).contiguous() if n_bands else cells_u.new_zeros((0, S)),
                    ids32.to(torch.uint16).contiguous(), zero_ptr.to(torch.int32).contiguous(),
                    zero_rows.to(torch.int32).contiguous(), idx_unit)
    sp.n_edges = E
    sp.row_bytes = row_bytes
    return sp


class CompactRows(object):
    """Node-major compact numbering of the (node, relation) pairs that have an edge (include/tipk.h section 2d):
    node_ptr [N + 1], row_rel [n_rows], pos [N, ceil(R / 64) * 64] (n_rows where a pair has no edge),
    node_desc [N, 4] = (node, first row, end row, 0) of the nodes by decreasing row count -- all int32 on the plan's device."""

    def __init__(self, n_rows, node_ptr, row_rel, pos, node_desc):
        self.n_rows, self.node_ptr, self.row_rel, self.pos, self.node_desc = int(n_rows), node_ptr, row_rel, pos, node_desc

    def to(self, device):
        return CompactRows(self.n_rows, self.node_ptr.to(device), self.row_rel.to(device), self.pos.to(device),
                           self.node_desc.to(device))


def compact_rows(out_node, rel, n_nodes, n_rel):
    """-> (CompactRows, row index of every edge).  Row of (node u, relation r) = node_ptr[u] + rank of r among the
    relations that have an edge at u (ascending)."""
    dev = out_node.device
    N, R = int(n_nodes), int(n_rel)
    r_pad = -(-R // 64) * 64
    has = torch.zeros(N * r_pad, dtype=torch.bool, device=dev)
    key = out_node * r_pad + rel
    has[key] = True
    used = torch.nonzero(has).flatten()                                    # ascending (node, relation)
    n_rows = int(used.numel())
    pos = torch.full((N * r_pad,), n_rows, dtype=torch.int64, device=dev)
    pos[used] = torch.arange(n_rows, device=dev)
    per_node = torch.bincount(used // r_pad, minlength=N)
    node_ptr = torch.cat([per_node.new_zeros(1), torch.cumsum(per_node, 0)])
    order = torch.sort(per_node, descending=True, stable=True).indices
    node_desc = torch.stack([order, node_ptr[order], node_ptr[order + 1], torch.zeros_like(order)], dim=1)
    cr = CompactRows(n_rows, node_ptr.to(torch.int32).contiguous(), (used % r_pad).to(torch.int32).contiguous(),
                     pos.view(N, r_pad).to(torch.int32).contiguous(), node_desc.to(torch.int32).contiguous())
    return cr, pos[key]


def build_stream_plan(out_node, tab_node, rel, n_nodes, n_rel, n_wg, lanes, piece=4, compact=False):
    """The (relation, node) form: out[row(r, o)] = sum_{e in r: out_node[e] = o} table[tab_node[e]], table =
    [N, d] -- the transposed D-D pass.
    compact = False: row(r, o) = r * N + o, all R N rows (those without edges through the zero-row list or `row_used`,
    the bit mask of the rows with edges); compact = True: only the rows WITH edges exist, numbered node-major
    (`compact_rows`; plan.compact holds the tables of tipk_rgcn_node_products)."""
    N = int(n_nodes)
    dev = out_node.device
    if compact:
        cr, row = compact_rows(out_node, rel, N, n_rel)
        sp = build_stream_plan_rows(row, tab_node, cr.n_rows, N, n_wg, lanes, piece)
        sp.compact = cr
        sp.n_nodes, sp.n_rel = N, int(n_rel)
        return sp
    sp = build_stream_plan_rows(rel * N + out_node, tab_node, n_rel * N, N, n_wg, lanes, piece)
    cnt_rows = torch.bincount(rel * N + out_node, minlength=n_rel * N)
    # bit (r & 31) of row_used[r >> 5, node] = row (r, node) has edges
    rt = -(-n_rel // 32)
    has = torch.zeros(rt * 32, N, dtype=torch.int64, device=dev)
    has[:n_rel] = (cnt_rows.view(n_rel, N) > 0).long()
    word = (has.view(rt, 32, N) << torch.arange(32, device=dev).view(1, 32, 1)).sum(1)
    sp.row_used = torch.where(word >= 2 ** 31, word - 2 ** 32, word).to(torch.int32).contiguous()
    sp.n_nodes, sp.n_rel = N, int(n_rel)
    return sp


def execute_stream_plan_reference(plan, table):
    """Pure-torch interpretation of a wave-stream plan (CPU unit tests only): what every slot of every
    wavefront does, in order; checks that every output row is written exactly once."""
    T, S, P = plan.n_table, 64 // plan.lanes, plan.piece
    d = table.shape[1]
    ids = plan.ids.to(torch.int64)
    assert bool((ids % plan.idx_unit == 0).all())
    ids = (ids // plan.idx_unit).view(-1, P, S, 8)
    cells = plan.cells.to(torch.int64) & 0xffffffff
    out = torch.zeros((plan.n_rows, d), dtype=table.dtype)
    written = torch.zeros(plan.n_rows, dtype=torch.long)
    wp, zp = plan.wave_ptr.tolist(), plan.zero_ptr.tolist()
    assert len(wp) == plan.n_wg * 16 + 1 and wp[0] == 0 and wp[-1] == plan.n_bands
    for w in range(plan.n_wg * 16):
        acc = torch.zeros((S, d), dtype=table.dtype)
        for b in range(wp[w], wp[w + 1]):
            klogs = [0] * S
            lasts = []
            for s_ in range(S):
                c = int(cells[b, s_])
                if c == 0:
                    continue
                row, ln, first, last, klogs[s_] = c & 0xffffff, (c >> 24) & 15, (c >> 28) & 1, (c >> 29) & 1, (c >> 30) & 3
                assert 0 <= ln <= P
                if first:
                    acc[s_] = 0
                nodes = ids[b, :ln, s_].reshape(-1)
                nodes = nodes[nodes < T]
                acc[s_] += table[nodes].sum(0)
                if last:
                    lasts.append((s_, row))
            if any(klogs):                                  # the kernel's tree: slot s receives slot s + 2^j when its set allows
                for jj in range(3):
                    for s_ in range(S):
                        if klogs[s_] > jj and s_ % (2 << jj) == 0 and s_ + (1 << jj) < S:
                            assert klogs[s_ + (1 << jj)] == klogs[s_], 'a set occupies aligned adjacent slots'
                            acc[s_] = acc[s_] + acc[s_ + (1 << jj)]
            for s_, row in lasts:
                out[row] = acc[s_]
                written[row] += 1
        for z in range(zp[w], zp[w + 1]):
            written[int(plan.zero_rows[z])] += 1
    assert bool((written == 1).all()), 'every output row is written exactly once'
    return out


def execute_rel_plan_reference(plan, table, backward):
    """Pure-torch interpretation of a relation-local plan (CPU unit tests only)."""
    n, R = plan.n_nodes, plan.n_rel
    d = table.shape[1]
    idx = plan.idx.to(torch.int64)
    assert bool((idx % plan.idx_unit == 0).all())
    idx = idx // plan.idx_unit
    runs = plan.runs.to(torch.int64)
    node_at = plan.node_at.to(torch.int64)                 # [U, N]
    res = torch.zeros((R, n, d), dtype=table.dtype)
    written = torch.zeros((R, n), dtype=torch.long)
    for u in range(plan.n_units):
        r = int(plan.unit_rel[u])
        e0 = int(plan.rel_idx_off[u])
        assert e0 % 8 == 0
        for p in range(int(plan.unit_npos[u])):
            b, ln = int(runs[u, p, 0]), int(runs[u, p, 1])
            written[r, node_at[u, p]] += 1
            if ln:
                assert b % 8 == 0 and ln % 8 == 0
                rows = idx[e0 + b:e0 + b + ln]
                rows = rows[rows < n]                                  # drop the padding sentinel
                src = table[rows] if backward else table[r * n + rows]
                res[r, node_at[u, p]] = src.sum(0)
        assert int(runs[u, int(plan.unit_npos[u]):, 1].sum()) == 0     # nothing beyond the walked positions
    if backward:
        assert bool((written == 1).all()), 'every (relation, node) row is written exactly once'
        return res.view(R * n, d)
    assert int(written.max()) <= 1
    return res.sum(0)


# ---------------------------------------------------------------------------------------------
# pair-form BACKWARD pass (include/tipk.h section 2e)
# ---------------------------------------------------------------------------------------------
PAIR_PART_ROWS = 1016       # most pairs of one partition of the d att gather: (rows + 1) x 128 B of LDS, 16-bit ids x 64
PAIR_PART_WGS = 4           # workgroups that share a partition (each stages it)
PAIR_PART_EDGES_PER_WG = 16384


class PairBwdPlan(object):
    """Device arrays of `tipk_rgcn_pair_grads` + `tipk_stream_gather_parts` (layout: include/tipk.h section 2e).

    slots [n_slots, 4] int32 {v, bits of 1 / deg(v), cell line, row of pg the slot's gradient row is written to},
    node_desc [N, 4] int32 {u, first slot, tiles, 0}; pg = two tables of n_alloc rows + one dump row: row t of the first
    table = the gradient row of pair t = (u, v), of the second = that of the mirrored pair (v, u) (symmetric graphs; zeros
    otherwise); part_first [n_parts] int32 = first pair of a partition, wg_part [n_wg] int32, and the merged wave-stream
    arrays of the partitions (`gather`: a StreamPlan whose rows are p * n_rel + r, ids = pairs counted from the partition's
    first)."""

    def __init__(self, n_nodes, n_rel, n_slots, slots, node_desc, n_parts, part_len, part_first, wg_part, gather, symmetric,
                 n_alloc, slot_of_pair=None):
        self.n_nodes, self.n_rel, self.n_slots = int(n_nodes), int(n_rel), int(n_slots)
        self.slots, self.node_desc = slots, node_desc
        # node of every tile of 32 slots (role 2 of the launch computes the pair-gradient rows one tile per wavefront)
        nd = node_desc.to(torch.int64)
        self.tile_node = torch.repeat_interleave(nd[:, 0], nd[:, 2])[torch.argsort(torch.repeat_interleave(nd[:, 1], nd[:, 2]), stable=True)].to(torch.int32).contiguous()
        self.n_parts, self.part_len, self.part_first, self.wg_part, self.gather = int(n_parts), int(part_len), part_first, wg_part, gather
        self.symmetric, self.n_alloc = bool(symmetric), int(n_alloc)
        self.slot_of_pair = slot_of_pair          # (tests) int64 [n_directed_pairs, 3] = (u, v, slot)
        self.pg = {}                              # device -> the pair-gradient buffer [2 * n_alloc + 1, n_bases], zeroed ONCE

    def to(self, device):
        mv = lambda t: None if t is None else t.to(device)
        return PairBwdPlan(self.n_nodes, self.n_rel, self.n_slots, mv(self.slots), mv(self.node_desc), self.n_parts,
                           self.part_len, mv(self.part_first), mv(self.wg_part), self.gather.to(device), self.symmetric,
                           self.n_alloc, mv(self.slot_of_pair))


def build_pair_bwd_plan(src, dst, rel, n_nodes, n_rel, scale, symmetric, n_wg=256, lanes=8, piece=4,
                        part_rows_max=PAIR_PART_ROWS, line_stride=None):
    """Plan of the pair-form backward pass of a D-D graph (src -> dst edges of relation rel; `scale` [N] = 1 / in-degree as
    the layer applies it).  symmetric: every relation links u -> v iff v -> u, and the forward pass kept the cells with
    u <= v only (`rgcn_graph`): a cell is read at line min * N + max, and the d att gather walks the edges with u <= v over
    the sums of the two gradient rows of a pair.  Otherwise: line u * N + v, all edges, one term.
    line_stride: nodes per row of the cell matrix (default n_nodes).

    The pairs the gather walks are dealt to partitions that fit in LDS, with equal numbers of pairs and of EDGES (a pair is
    linked by 1 ... 475 relations at BioSNAP); PAIR_PART_WGS workgroups share a partition, the relations' runs inside it are
    dealt to their wavefronts by `build_stream_plan_rows`."""
    dev = src.device
    N, R = int(n_nodes), int(n_rel)
    ls = N if line_stride is None else int(line_stride)
    src, dst, rel = src.to(torch.int64), dst.to(torch.int64), rel.to(torch.int64)
    # ---- directed pairs, grouped by source node, neighbours ascending; 32 slots per tile
    key = torch.unique(src * N + dst)                                    # sorted
    pu, pv = key // N, key % N
    n_dp = int(key.numel())
    deg = torch.bincount(pu, minlength=N)
    tiles = (deg + 31) // 32
    first_tile = torch.cumsum(tiles, 0) - tiles
    n_slots = int(tiles.sum()) * 32
    first_pair = torch.cumsum(deg, 0) - deg
    slot = first_tile[pu] * 32 + (torch.arange(n_dp, device=dev) - first_pair[pu])
    line = torch.where(pu <= pv, pu * ls + pv, pv * ls + pu) if symmetric else pu * ls + pv
    assert n_slots > 0 and int(line.max()) < 2 ** 24
    # ---- the table of the d att gather: one row per pair walked
    if symmetric:
        keep = pu <= pv
        t_key = key[keep]                                                 # ascending (u, v), u <= v
        ek = src <= dst
        e_key, e_rel = (src * N + dst)[ek], rel[ek]
        lo, hi = torch.minimum(pu, pv), torch.maximum(pu, pv)
        row_of = torch.searchsorted(t_key, lo * N + hi)                   # pair (u, v) and its mirror share a row
        assert bool((t_key[row_of.clamp(max=t_key.numel() - 1)] == lo * N + hi).all()), 'graph is not symmetric'
    else:
        t_key = key
        e_key, e_rel = src * N + dst, rel
        row_of = torch.arange(n_dp, device=dev)
    n_t = int(t_key.numel())
    E = int(e_key.numel())
    e_row = torch.searchsorted(t_key, e_key)                              # table row of every edge walked
    # ---- partitions: equal numbers of rows AND of edges.  Where a pair's row sits in the table is free (the slots carry it),
    # so the pairs are dealt to the partitions like cards, heaviest first, back and forth (BioSNAP: 1 ... 475 relations per
    # pair, and 43 k ... 89 k edges per 1 000 consecutive pairs: consecutive blocks were 20 % apart)
    cap = int(part_rows_max)
    assert (cap + 1) * lanes * 16 <= 158 * 1024
    n_parts = max(1, -(-n_t // cap))
    if E >= 2 * PAIR_PART_WGS * PAIR_PART_EDGES_PER_WG:
        n_parts = max(n_parts, min(int(n_wg) // PAIR_PART_WGS, -(-E // (PAIR_PART_WGS * PAIR_PART_EDGES_PER_WG))))
    part_len = -(-(-(-n_t // n_parts)) // 8) * 8
    row_edges = torch.bincount(e_row, minlength=n_t)
    by_load = torch.sort(row_edges, descending=True, stable=True).indices
    i = torch.arange(n_t, device=dev)
    k, j = i // n_parts, i % n_parts
    part_of = torch.empty(n_t, dtype=torch.int64, device=dev)
    part_of[by_load] = torch.where(k % 2 == 0, j, n_parts - 1 - j)
    # inside a partition the pairs keep their (u, v) order: a node's gradient rows land near each other
    o_p = torch.sort(part_of, stable=True).indices
    p_cnt = torch.bincount(part_of, minlength=n_parts)
    p_start = torch.cumsum(p_cnt, 0) - p_cnt
    new_row = torch.empty(n_t, dtype=torch.int64, device=dev)
    new_row[o_p] = part_of[o_p] * part_len + (torch.arange(n_t, device=dev) - p_start[part_of[o_p]])
    e_row, row_of = new_row[e_row], new_row[row_of]
    part_first = torch.arange(n_parts, device=dev) * part_len
    n_alloc = n_parts * part_len
    e_part, e_local = e_row // part_len, e_row % part_len
    # ---- slots: {v, 1 / deg(v), cell line, destination row of the gradient row}
    # pads: the node's own first neighbour / line with the factor 0 (they add zeros); their gradient rows go to the dump row
    slot_node = torch.repeat_interleave(torch.arange(N, device=dev), tiles * 32)
    fp = first_pair.clamp(max=max(n_dp - 1, 0))
    sl_v, sl_line = pv[fp][slot_node].clone(), line[fp][slot_node].clone()
    sl_scale = torch.zeros(n_slots, dtype=torch.float32, device=dev)
    sl_dest = torch.full((n_slots,), 2 * n_alloc, dtype=torch.int64, device=dev)
    sl_v[slot], sl_line[slot] = pv, line
    sl_scale[slot] = scale.to(dev).to(torch.float32)[pv]
    sl_dest[slot] = torch.where(pu <= pv, row_of, n_alloc + row_of) if symmetric else row_of
    assert 2 * n_alloc + 1 < 2 ** 25
    slots = torch.stack([sl_v.to(torch.int32), sl_scale.view(torch.int32), sl_line.to(torch.int32), sl_dest.to(torch.int32)],
                        dim=1).contiguous()
    order = torch.sort(tiles, descending=True, stable=True).indices
    node_desc = torch.stack([order, first_tile[order] * 32, tiles[order], torch.zeros_like(order)], dim=1).to(torch.int32).contiguous()
    # ---- workgroups per partition
    e_cnt = torch.bincount(e_part, minlength=n_parts).cpu().tolist()
    per = max(1, min(int(n_wg) // n_parts, -(-E // (n_parts * PAIR_PART_EDGES_PER_WG)) if E else 1))
    wgs = [per] * n_parts                                                 # (equal loads: equal shares)
    n_wg = sum(wgs)
    # ---- one wave-stream plan per partition, cut into its workgroups' pieces ...
    S = 64 // lanes
    pieces = {}                                                           # (partition, workgroup of it) -> its arrays
    idx_unit = None
    o_part = torch.sort(e_part, stable=True).indices
    p_first = [0]
    for c in e_cnt:
        p_first.append(p_first[-1] + c)
    for p in range(n_parts):
        sel = o_part[p_first[p]:p_first[p + 1]]
        sp = build_stream_plan_rows(e_rel[sel], e_local[sel], R, part_len, wgs[p], lanes, piece)
        assert idx_unit in (None, sp.idx_unit)
        idx_unit = sp.idx_unit
        c = sp.cells.to(torch.int64) & 0xffffffff
        c = torch.where(c != 0, c + p * R, c)                             # output row = p * R + relation (24-bit field)
        c = torch.where(c >= 2 ** 31, c - 2 ** 32, c).to(torch.int32)
        wp, zp = sp.wave_ptr.to(torch.int64), sp.zero_ptr.to(torch.int64)
        idw = sp.ids.view(-1, piece * S * 8)                              # one row per band
        for q in range(wgs[p]):
            b0, b1 = int(wp[16 * q]), int(wp[16 * q + 16])
            y0, y1 = int(zp[16 * q]), int(zp[16 * q + 16])
            pieces[(p, q)] = (wp[16 * q:16 * q + 16] - b0, c[b0:b1], idw[b0:b1], zp[16 * q:16 * q + 16] - y0,
                              sp.zero_rows[y0:y1].to(torch.int64) + p * R)
    # ... laid out in LAUNCH order: workgroup b runs on XCD b mod 8, and the workgroups that stage the same partition should
    # share an XCD -- then its rows cross the fabric once and the other three read them out of that XCD's L2 (in partition
    # order the four sat on four XCDs: 66 MB instead of 17 MB out of the Infinity Cache per launch)
    n_xcd = 8
    per_xcd = [[(p, q) for p in range(x, n_parts, n_xcd) for q in range(wgs[p])] for x in range(n_xcd)]
    order = []
    while any(per_xcd):
        for x in range(n_xcd):
            src_l = per_xcd[x] if per_xcd[x] else max(per_xcd, key=len)
            if src_l:
                order.append(src_l.pop(0))
    wave_ptr, cells, ids, zero_ptr, zero_rows, wg_part = [], [], [], [], [], []
    band0 = z0 = 0
    for (p, q) in order:
        w_, c_, i_, z_, zr_ = pieces[(p, q)]
        wave_ptr.append(w_ + band0)
        zero_ptr.append(z_ + z0)
        cells.append(c_)
        ids.append(i_.reshape(-1))
        zero_rows.append(zr_)
        band0 += int(c_.shape[0])
        z0 += int(zr_.numel())
        wg_part.append(p)
    assert n_parts * R < 2 ** 24
    gather = StreamPlan(n_parts * R, part_len, n_wg, lanes, piece,
                        torch.cat(wave_ptr + [torch.tensor([band0], device=dev)]).to(torch.int32).contiguous(),
                        torch.cat(cells).view(-1, S).contiguous(),
                        (torch.cat(ids) if band0 else torch.full((piece * S * 8,), part_len * idx_unit, dtype=torch.int32, device=dev).to(torch.uint16)).contiguous(),
                        torch.cat(zero_ptr + [torch.tensor([z0], device=dev)]).to(torch.int32).contiguous(),
                        (torch.cat(zero_rows) if z0 else torch.zeros(1, dtype=torch.int64, device=dev)).to(torch.int32).contiguous(),
                        idx_unit)
    gather.n_edges = E
    return PairBwdPlan(N, R, n_slots, slots, node_desc, n_parts, part_len, part_first.to(torch.int32).contiguous(),
                       torch.tensor(wg_part, dtype=torch.int32, device=dev), gather, symmetric, n_alloc,
                       torch.stack([pu, pv, slot], dim=1))


def execute_pair_bwd_reference(plan, cells_flat, xb, g, n_bases):
    """Pure-torch interpretation of a pair-backward plan (CPU unit tests): cells_flat [n_lines, n_bases], xb [N, n_bases, d],
    g [N, d] -> (dxb [n_bases, N, d], pg [2 * n_alloc + 1, n_bases], datt [n_rel, n_bases]) exactly as the two kernels sum them
    (up to the order inside a tile)."""
    N, d = g.shape
    sl = plan.slots.to(torch.int64)
    v, line, dest = sl[:, 0], sl[:, 2], sl[:, 3]
    sc = plan.slots[:, 1].contiguous().view(torch.float32).to(g.dtype)
    gp = g[v] * sc.unsqueeze(1)                                           # [n_slots, d]
    nd = plan.node_desc.to(torch.int64)
    node_of_slot = torch.empty(plan.n_slots, dtype=torch.int64)
    for u, s0, nt, _ in nd.tolist():
        node_of_slot[s0:s0 + 32 * nt] = u
    dxb = torch.zeros((N, n_bases, d), dtype=g.dtype)
    dxb.index_add_(0, node_of_slot, cells_flat[line].unsqueeze(2) * gp.unsqueeze(1))
    pg = torch.zeros((2 * plan.n_alloc + 1, n_bases), dtype=g.dtype)
    real = dest < 2 * plan.n_alloc
    assert int(torch.unique(dest[real]).numel()) == int(real.sum()), 'every gradient row has a place of its own'
    pg[dest] = torch.einsum('sbc,sc->sb', xb[node_of_slot], gp)           # (pads all land in the dump row)
    pg[2 * plan.n_alloc] = 0
    both = pg[:plan.n_alloc] + pg[plan.n_alloc:2 * plan.n_alloc]
    # the gather, partition by partition, on the merged stream plan
    gp_ = plan.gather
    S, P = 64 // gp_.lanes, gp_.piece
    ids = (gp_.ids.to(torch.int64) // gp_.idx_unit).view(-1, P, S, 8)
    cw = gp_.cells.to(torch.int64) & 0xffffffff
    out = torch.zeros((gp_.n_rows, n_bases), dtype=g.dtype)
    written = torch.zeros(gp_.n_rows, dtype=torch.long)
    wp, zp = gp_.wave_ptr.tolist(), gp_.zero_ptr.tolist()
    wg_part, part_first = plan.wg_part.tolist(), plan.part_first.tolist()
    assert len(wp) == gp_.n_wg * 16 + 1 and wp[-1] == gp_.n_bands
    for w in range(gp_.n_wg * 16):
        r0 = part_first[wg_part[w // 16]]
        tab = torch.cat([both[r0:r0 + plan.part_len], torch.zeros((1, n_bases), dtype=g.dtype)])
        acc = torch.zeros((S, n_bases), dtype=g.dtype)
        for b in range(wp[w], wp[w + 1]):
            klogs, lasts = [0] * S, []
            for s_ in range(S):
                c = int(cw[b, s_])
                if c == 0:
                    continue
                row, ln, first, last, klogs[s_] = c & 0xffffff, (c >> 24) & 15, (c >> 28) & 1, (c >> 29) & 1, (c >> 30) & 3
                if first:
                    acc[s_] = 0
                rows = ids[b, :ln, s_].reshape(-1)
                acc[s_] += tab[rows].sum(0)
                if last:
                    lasts.append((s_, row))
            if any(klogs):
                for jj in range(3):
                    for s_ in range(S):
                        if klogs[s_] > jj and s_ % (2 << jj) == 0 and s_ + (1 << jj) < S:
                            acc[s_] = acc[s_] + acc[s_ + (1 << jj)]
            for s_, row in lasts:
                assert row // plan.n_rel == wg_part[w // 16], 'a wavefront writes rows of its own partition only'
                out[row] = acc[s_]
                written[row] += 1
        for z in range(zp[w], zp[w + 1]):
            written[int(gp_.zero_rows[z])] += 1
    assert bool((written == 1).all()), 'every (partition, relation) row is written exactly once'
    datt = out.view(plan.n_parts, plan.n_rel, n_bases).sum(0)
    return dxb.permute(1, 0, 2).contiguous(), pg, datt


# ---------------------------------------------------------------------------------------------
# destination-major edge list (include/tipk.h section 2f)
# ---------------------------------------------------------------------------------------------
class DestPlan(object):
    """edges [E] int32 (uint32 words rel | src << bits) grouped by destination, node_desc [N, 4] int32 {v, first, count, 0} by
    decreasing count."""

    def __init__(self, n_nodes, n_rel, bits, edges, node_desc):
        self.n_nodes, self.n_rel, self.bits, self.edges, self.node_desc = int(n_nodes), int(n_rel), int(bits), edges, node_desc
        self.n_edges = int(edges.numel())

    def to(self, device):
        return DestPlan(self.n_nodes, self.n_rel, self.bits, self.edges.to(device), self.node_desc.to(device))


def build_dest_plan(src, dst, rel, n_nodes, n_rel, bits):
    """Edges sorted by destination (stable: relation-major order inside a node, as the edge list has it)."""
    dev = src.device
    src, dst, rel = src.to(torch.int64), dst.to(torch.int64), rel.to(torch.int64)
    assert int(n_rel) <= (1 << bits) and int(n_nodes) <= (1 << (32 - bits))
    order = torch.sort(dst, stable=True).indices
    w = rel[order] | (src[order] << bits)
    w = torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32).contiguous()
    cnt = torch.bincount(dst, minlength=int(n_nodes))
    first = torch.cumsum(cnt, 0) - cnt
    by = torch.sort(cnt, descending=True, stable=True).indices
    desc = torch.stack([by, first[by], cnt[by], torch.zeros_like(by)], dim=1).to(torch.int32).contiguous()
    if w.numel() == 0:
        w = torch.zeros(1, dtype=torch.int32, device=dev)
    return DestPlan(n_nodes, n_rel, bits, w, desc)


class RowStreamPlan(object):
    """Per (node, relation tile of 32) the edges of the node as batches of 16 entry words per lane half
    (`tipk_rgcn_row_products`, include/tipk.h section 2h):

        entries [n_batches, 2, 16] int32   word = inside << 24 | other << 8 | 4 * (rel % 32), sorted by row inside a half's
                                           list; inside = 0 at the first entry of a (relation, node) row, 1 at the others;
                                           padding = 128 (node 0 into the dump column), only at the end of a list; the two
                                           halves of a (node, tile) hold disjoint sets of rows
        desc    [n_nodes, n_tiles, 2] int32  {first batch, batches >= 1} of the tile; a node's batches are consecutive, and
                                           the array ends with 8 batches of padding (the kernel's loads run 8 batches ahead)

    `key` is the node that owns a row (destination for the forward pass, source for the transposed pass), `other` the
    node whose table row is gathered."""

    def __init__(self, n_nodes, n_rel, entries, desc, n_edges):
        self.n_nodes, self.n_rel, self.entries, self.desc, self.n_edges = int(n_nodes), int(n_rel), entries, desc, int(n_edges)
        self.n_tiles = (self.n_rel + 31) // 32

    def to(self, device):
        return RowStreamPlan(self.n_nodes, self.n_rel, self.entries.to(device), self.desc.to(device), self.n_edges)


def build_row_stream_plan(key, other, rel, n_nodes, n_rel):
    """Edges sorted by (key, relation) -- stable, so equal rows keep the edge list's order and the sums are a fixed
    sequence.  The sorted list of a (node, tile) is cut in two at the row boundary nearest its middle: the first part is
    walked by lanes 0-31, the second by lanes 32-63 (the rows of a tile meet in LDS, so any split by whole rows will do)."""
    dev = key.device
    n_nodes, n_rel = int(n_nodes), int(n_rel)
    n_tiles = (n_rel + 31) // 32
    key, other, rel = key.to(torch.int64), other.to(torch.int64), rel.to(torch.int64)
    assert n_nodes <= (1 << 16)
    e = int(key.numel())
    n_seg = n_nodes * n_tiles
    seg = key * n_tiles + rel // 32                                     # (node, tile) segment of every edge
    row_key = key * n_rel + rel
    order = torch.sort(row_key, stable=True).indices                   # (node, relation): segments in (node, tile) order
    seg_s, row_s = seg[order], row_key[order]
    n_in = torch.bincount(seg_s, minlength=n_seg)
    cut = torch.zeros(n_seg, dtype=torch.int64, device=dev)
    if e:
        seg_first = torch.cumsum(n_in, 0) - n_in
        idx = torch.arange(e, device=dev)
        pos = idx - seg_first[seg_s]                                   # rank inside the segment's list
        starts = torch.ones(e, dtype=torch.bool, device=dev)
        starts[1:] = row_s[1:] != row_s[:-1]
        row_start = pos[torch.cummax(torch.where(starts, idx, torch.zeros_like(idx)), 0).values]   # rank of the row's first entry
        # the row start s that minimises max(s, n - s): the longer part decides the number of batches
        cost = torch.maximum(row_start, n_in[seg_s] - row_start) * (1 << 20) + row_start
        best = torch.full((n_seg,), (1 << 62), dtype=torch.int64, device=dev)
        best.scatter_reduce_(0, seg_s[starts], cost[starts], 'amin', include_self=True)
        cut = torch.where(n_in > 0, best % (1 << 20), torch.zeros_like(best))
        half = (row_start >= cut[seg_s]).to(torch.int64)
        rank = pos - half * cut[seg_s]
    longer = torch.maximum(cut, n_in - cut)
    nbat = torch.clamp((longer + 15) // 16, min=1)                     # every tile at least one batch (the kernel's load
    first = torch.cumsum(nbat, 0) - nbat                               # pipeline never branches on a tile's length)
    n_batches = int(nbat.sum())
    entries = torch.full((n_batches + 8, 2, 16), 128, dtype=torch.int32, device=dev)   # + padding the pipeline runs into
    if e:
        inside = (~starts).to(torch.int64)
        word = (inside << 24) | (other[order] << 8) | ((rel[order] % 32) * 4)
        flat = ((first[seg_s] + rank // 16) * 2 + half) * 16 + rank % 16
        entries.view(-1)[flat] = word.to(torch.int32)
    desc = torch.stack([first, nbat], dim=1).to(torch.int32).view(n_nodes, n_tiles, 2).contiguous()
    return RowStreamPlan(n_nodes, n_rel, entries.contiguous(), desc, e)


class RowStreamPlanS(object):
    """Wave-uniform form of the row-sum plan (`tipk_rgcn_row_products_s`, include/tipk.h section 2h): ONE list per (node, tile
    of 32 relations), sorted by relation, padded to batches of 16 entries:

        entries [n_batches, 2, 16] int32   plane 0 = `other` (the node whose table row is gathered; `entries_for(ld_bytes)`
                                           multiplies it by the bytes of a table row: what the kernel's buffer loads take as
                                           their scalar offset), plane 1 = 260 * row | 0x3f800000 inside a (relation, node)
                                           row (260 = 4 * 65: the byte stride of the kernel's LDS tile; row 32 = padding)
        desc    [n_nodes, n_tiles, 2] int32  {first batch, batches >= 1}; the array ends with 8 batches of padding"""
    ROW_BYTES = 260

    def __init__(self, n_nodes, n_rel, entries, desc, n_edges):
        self.n_nodes, self.n_rel, self.entries, self.desc, self.n_edges = int(n_nodes), int(n_rel), entries, desc, int(n_edges)
        self.n_tiles = (self.n_rel + 31) // 32
        self._by_ld = {}

    def to(self, device):
        return RowStreamPlanS(self.n_nodes, self.n_rel, self.entries.to(device), self.desc.to(device), self.n_edges)

    def entries_for(self, ld_bytes):
        ld_bytes = int(ld_bytes)
        if ld_bytes not in self._by_ld:
            assert self.n_nodes * ld_bytes < 2 ** 31
            e = self.entries.clone()
            e[:, 0, :] *= ld_bytes
            self._by_ld[ld_bytes] = e.contiguous()
        return self._by_ld[ld_bytes]


def build_row_stream_plan_s(key, other, rel, n_nodes, n_rel):
    """Edges sorted by (key, relation), stable: the sums are a fixed sequence."""
    dev = key.device
    n_nodes, n_rel = int(n_nodes), int(n_rel)
    n_tiles = (n_rel + 31) // 32
    key, other, rel = key.to(torch.int64), other.to(torch.int64), rel.to(torch.int64)
    e = int(key.numel())
    n_seg = n_nodes * n_tiles
    seg = key * n_tiles + rel // 32
    row_key = key * n_rel + rel
    order = torch.sort(row_key, stable=True).indices
    seg_s, row_s = seg[order], row_key[order]
    n_in = torch.bincount(seg_s, minlength=n_seg)
    nbat = torch.clamp((n_in + 15) // 16, min=1)
    first = torch.cumsum(nbat, 0) - nbat
    n_batches = int(nbat.sum())
    entries = torch.zeros((n_batches + 8, 2, 16), dtype=torch.int32, device=dev)
    entries[:, 1, :] = 32 * RowStreamPlanS.ROW_BYTES                     # padding: row 0 of the table into the dump row
    if e:
        seg_first = torch.cumsum(n_in, 0) - n_in
        pos = torch.arange(e, device=dev) - seg_first[seg_s]
        inside = torch.zeros(e, dtype=torch.int64, device=dev)
        inside[1:] = (row_s[1:] == row_s[:-1]).to(torch.int64)
        flat0 = (first[seg_s] + pos // 16) * 32 + pos % 16
        entries.view(-1)[flat0] = other[order].to(torch.int32)
        entries.view(-1)[flat0 + 16] = ((rel[order] % 32) * RowStreamPlanS.ROW_BYTES + inside * 0x3f800000).to(torch.int32)
    desc = torch.stack([first, nbat], dim=1).to(torch.int32).view(n_nodes, n_tiles, 2).contiguous()
    return RowStreamPlanS(n_nodes, n_rel, entries.contiguous(), desc, e)


def execute_row_stream_s_reference(plan, table, att, xb=None):
    """What `tipk_rgcn_row_products_s` computes, from the plan alone (float64)."""
    n, r, nt = plan.n_nodes, plan.n_rel, plan.n_tiles
    table, att = table.double().cpu(), att.double().cpu()
    ent = plan.entries.cpu().to(torch.int64)
    desc = plan.desc.cpu().to(torch.int64)
    ch = table.shape[1]
    s = torch.zeros((nt * 32 + 1, n, ch), dtype=torch.float64)
    for v in range(n):
        for tl in range(nt):
            f, nb_ = int(desc[v, tl, 0]), int(desc[v, tl, 1])
            oth = ent[f:f + nb_, 0, :].reshape(-1)
            w1 = ent[f:f + nb_, 1, :].reshape(-1)
            row = (w1 & 0xffff) // RowStreamPlanS.ROW_BYTES
            rows = torch.where(row < 32, tl * 32 + row, torch.full_like(row, nt * 32))
            s[:, v].index_add_(0, rows, table[oth])
    s = s[:r]
    t = torch.einsum('rb,rvc->bvc', att, s)
    if xb is None:
        return t
    return t, torch.einsum('rvc,bvc->rb', s, xb.double().cpu())


def execute_row_stream_reference(plan, table, att, xb=None):
    """What `tipk_rgcn_row_products` computes, from the plan alone (float64): T [nb, n, ch] and (with xb [nb, n, ch]) d att."""
    n, r, nt = plan.n_nodes, plan.n_rel, plan.n_tiles
    table = table.double().cpu()
    att = att.double().cpu()
    ent = plan.entries.cpu().to(torch.int64)
    desc = plan.desc.cpu().to(torch.int64)
    ch = table.shape[1]
    s = torch.zeros((nt * 32 + 1, n, ch), dtype=torch.float64)          # row r of node v; the last row collects padding
    for v in range(n):
        for tl in range(nt):
            f, nb_ = int(desc[v, tl, 0]), int(desc[v, tl, 1])
            if nb_ == 0:
                continue
            w = ent[f:f + nb_].permute(1, 0, 2).reshape(2, -1)          # [half, entries]
            for h in range(2):
                oth, off = (w[h] >> 8) & 0xffff, (w[h] & 0xff) // 4
                rows = torch.where(off < 32, tl * 32 + off, torch.full_like(off, nt * 32))
                s[:, v].index_add_(0, rows, table[oth])
    s = s[:r]
    t = torch.einsum('rb,rvc->bvc', att, s)
    if xb is None:
        return t
    return t, torch.einsum('rvc,bvc->rb', s, xb.double().cpu())
