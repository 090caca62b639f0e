"""Python face of libtipk: thin launch wrappers + the `torch.autograd.Function`s of the TIP layers.

torch supplies device memory, the current HIP stream and autograd bookkeeping; every FLOP of the
path is executed by the HIP kernels in `tip_amd/csrc` through the C ABI (`include/tipk.h`).
Backward passes are explicit (no autograd tracing through E x d tensors): each Function saves
only N x d activations and the static graph plans.
"""
import os

import torch

from . import _lib, switches
from ._lib import SlabSumDesc, GemmDesc, WgGemmDesc, check, lib, ptr, stream_ptr, require_device

def _f32c(t):
    """fp32, unit column stride (row stride free)."""
    if t.dtype != torch.float32:
        raise _lib.TipkError('tip_amd computes in fp32; got %s' % t.dtype)
    if t.dim() >= 1 and t.stride(-1) != 1 and t.shape[-1] != 1:
        t = t.contiguous()
    return t


# ---------------------------------------------------------------------------------------------
# optional per-kernel timing (bench.py): HIP events recorded on the launch stream around one launch
# ---------------------------------------------------------------------------------------------
_TIMING = None          # None (off) or dict: label -> list of (start_event, end_event)


def timing_start():
    global _TIMING
    _TIMING = {}


def timing_stop():
    """-> {label: (n_launches, mean_ms)}; call after a device synchronize."""
    global _TIMING
    rec, _TIMING = _TIMING or {}, None
    return {k: (len(v), sum(a.elapsed_time(b) for a, b in v) / len(v)) for k, v in rec.items()}


class _timed(object):
    def __init__(self, label):
        self.label = label

    def __enter__(self):
        if _TIMING is not None:
            self.ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self.ev[0].record()

    def __exit__(self, *exc):
        if _TIMING is not None:
            self.ev[1].record()
            _TIMING.setdefault(self.label, []).append(self.ev)


# ---------------------------------------------------------------------------------------------
# launch wrappers (one C call each)
# ---------------------------------------------------------------------------------------------
def gather_sum_epilogue_supported(plan, d):
    """True if `gather_sum(..., gate=, colsum=True, riders=)` runs as ONE launch on this plan (grouped, 1024-thread workgroups)."""
    return bool(not plan.n_slots and plan.items.shape[0] > 0 and d % 4 == 0 and lib().tipk_gather_sum_riders_supported(d, plan.group_slots))


def gather_sum(plan, table, row_scale=None, bias=None, relu=False, out=None, riders=(), gate=None, colsum=False):
    """out[n_out, d] per `plan` over `table` [n_table, d] (include/tipk.h section 1).
    riders: up to 3 `slab_job`s that are ready now -- summed by further workgroups of the SAME launch when the plan's
    workgroups have 1024 threads (`tipk_gather_sum_riders`), by a grouped slab-sum launch of their own otherwise.
    gate [n_out, d]: out = gate > 0 ? out : 0;  colsum: also return the per-workgroup column sums of out as [W, 1, d] (both only
    where `gather_sum_epilogue_supported`)."""
    table = _f32c(table)
    require_device(table, plan.items)
    d = table.shape[1]
    assert table.shape[0] == plan.n_table, (table.shape, plan.n_table)
    if out is None:
        out = torch.empty((plan.n_out, d), dtype=torch.float32, device=table.device)
    riders = list(riders)
    st = stream_ptr(table.device)
    L = lib()
    one_launch = (riders or gate is not None or colsum) and len(riders) <= 3 and gather_sum_epilogue_supported(plan, d)
    assert one_launch or (gate is None and not colsum), 'gate / colsum need a grouped plan with 1024-thread workgroups'
    if one_launch:
        arr = (SlabSumDesc * max(1, len(riders)))(*[r.desc for r in riders])
        parts = None
        if colsum:
            parts = torch.empty((-(-plan.items.shape[0] // plan.group_slots), 1, d), dtype=torch.float32, device=table.device)
        if gate is not None:
            assert tuple(gate.shape) == tuple(out.shape) and gate.stride(1) == 1 and gate.dtype == torch.float32
        with _timed('gather_sum[%s,d=%d]+%d sums' % (plan.tag, d, len(riders))):
            check(L.tipk_gather_sum_riders(ptr(table), table.stride(0), table.shape[0], ptr(plan.row_id), ptr(plan.edge_w),
                                           ptr(plan.items), plan.items.shape[0], ptr(out), out.stride(0), ptr(row_scale),
                                           ptr(bias), int(relu), d, plan.group_slots, ptr(gate),
                                           gate.stride(0) if gate is not None else 0, ptr(parts), arr, len(riders), st),
                  'tipk_gather_sum_riders')
        return (out, parts) if colsum else out
    partial = None
    if plan.n_slots:
        partial = torch.empty((plan.n_slots, d), dtype=torch.float32, device=table.device)
    with _timed('gather_sum[%s,d=%d]' % (plan.tag, d)):
        check(L.tipk_gather_sum(ptr(table), table.stride(0), table.shape[0], ptr(plan.row_id), ptr(plan.edge_w), ptr(plan.items),
                                plan.items.shape[0], ptr(out), out.stride(0), ptr(partial), ptr(row_scale),
                                ptr(bias), int(relu), d, plan.group_slots, st), 'tipk_gather_sum')
    gather_sum_finish(plan, partial, out, row_scale, bias, relu)
    if riders:
        gemm_group([], riders)
    return out


def gather_sum_finish(plan, partial, out, row_scale=None, bias=None, relu=False):
    """Adds the pieces of the rows that a plan cut into several work items (slot order: deterministic)."""
    if plan.n_slots:
        d = out.shape[1]
        check(lib().tipk_gather_sum_finalize(ptr(partial), ptr(plan.split_rows), plan.split_rows.shape[0], ptr(out),
                                             out.stride(0), ptr(row_scale), ptr(bias), int(relu), d, plan.max_slots,
                                             stream_ptr(out.device)), 'tipk_gather_sum_finalize')


def gather_sum_lin_supported(d, d2, group_slots):
    return bool(lib().tipk_gather_sum_lin_supported(int(d), int(d2), int(group_slots)))


def gather_sum_lin(plan, table, weight, bias=None, relu=False, row_scale=None):
    """(agg, out2): agg = the plan's row sums over `table` [n_table, d] (x row_scale), out2 = relu?(agg @ weight^T + bias) for
    weight [d2, d] (any strides), in ONE launch on a grouped plan (include/tipk.h `tipk_gather_sum_lin`)."""
    table = _f32c(table)
    require_device(table, plan.items, weight)
    d, d2 = table.shape[1], weight.shape[0]
    assert table.shape[0] == plan.n_table and weight.shape[1] == d and plan.group_slots > 0 and not plan.n_slots
    agg = torch.empty((plan.n_out, d), dtype=torch.float32, device=table.device)
    out2 = torch.empty((plan.n_out, d2), dtype=torch.float32, device=table.device)
    with _timed('gather_sum_lin[%s,d=%d->%d]' % (plan.tag, d, d2)):
        check(lib().tipk_gather_sum_lin(ptr(table), table.stride(0), table.shape[0], ptr(plan.row_id), ptr(plan.edge_w),
                                        ptr(plan.items), plan.items.shape[0], ptr(agg), agg.stride(0), ptr(row_scale),
                                        ptr(weight), weight.stride(0), weight.stride(1), ptr(bias), int(relu), ptr(out2),
                                        out2.stride(0), d, d2, plan.group_slots, stream_ptr(table.device)), 'tipk_gather_sum_lin')
    return agg, out2


def gather_rows_csr(plan, table):
    """out[n_out, d] = per-row sums over a CsrPlan (include/tipk.h section 1c)."""
    table = _f32c(table)
    require_device(table, plan.row_ptr)
    d = table.shape[1]
    assert table.shape[0] == plan.n_table, (table.shape, plan.n_table)
    out = torch.empty((plan.n_out, d), dtype=torch.float32, device=table.device)
    with _timed('gather_rows_csr[%s,d=%d]' % (plan.tag, d)):
        check(lib().tipk_gather_rows_csr(ptr(table), table.stride(0), table.shape[0], ptr(plan.row_ptr), ptr(plan.row_id), plan.n_out,
                                         ptr(out), out.stride(0), d, stream_ptr(table.device)), 'tipk_gather_rows_csr')
    return out


def sum_slabs(slabs, out=None, alpha=1.0, accumulate=False, row_scale=None, addend=None, relu=False):
    """out = relu?(alpha * row_scale (.) sum_s slabs[s] + addend (+ out)): ordered, deterministic
    (include/tipk.h section 2).  slabs: [S, rows, cols]."""
    assert slabs.is_contiguous()
    n = slabs.shape[0]
    per = slabs[0].numel()
    if out is None:
        out = torch.empty(slabs.shape[1:], dtype=torch.float32, device=slabs.device)
    assert out.is_contiguous() and out.numel() == per
    if addend is not None:
        assert addend.is_contiguous() and addend.numel() == per
    with _timed('sum_slabs[%dx%d]' % (n, per)):
        check(lib().tipk_sum_slabs_ex(ptr(slabs), n, per, per, alpha, int(accumulate), ptr(row_scale), slabs.shape[-1],
                                      ptr(addend), int(relu), ptr(out), stream_ptr(slabs.device)), 'tipk_sum_slabs_ex')
    return out


def rel_gather_split(n_nodes, d, backward):
    """column blocks `tipk_rel_gather` would use for this shape; 0 = use the generic gather_sum."""
    if switches.on('TIPK_NO_RELLOCAL'):
        return 0
    return int(lib().tipk_rel_gather_supported(n_nodes, d, int(backward)))


def rel_gather_wgs(n_nodes, d, backward, n_cu):
    """workgroups per column block the relation-local plan of this shape should be built for:
    occupancy (1 or 2 workgroups per CU) * CUs / column blocks."""
    split = rel_gather_split(n_nodes, d, backward)
    if split == 0:
        return 0
    occ = int(lib().tipk_rel_gather_occupancy(n_nodes, d, int(backward)))
    return max(1, occ * n_cu // split)


def rel_gather_chunk(n_nodes, d, backward):
    return int(lib().tipk_rel_gather_chunk(n_nodes, d, int(backward)))


def rel_gather_usable(rp, n_nodes, d, backward):
    return rp is not None and rel_gather_split(n_nodes, d, backward) > 0


def rel_gather(rp, table, backward, row_scale=None, reduce=True):
    """LDS-resident D-D aggregation (include/tipk.h section 1b).  forward: table = Y [R*N, d] ->
    partial slabs [n_wg, N, d] (summed here unless reduce=False); backward: table = g [N, d]
    (optionally scaled per row while staged) -> dY [R*N, d]."""
    table = _f32c(table)
    require_device(table, rp.idx)
    d = table.shape[1]
    n, r = rp.n_nodes, rp.n_rel
    if backward:
        out = torch.empty((r * n, d), dtype=torch.float32, device=table.device)
    else:
        out = torch.empty((rp.n_wg, n, d), dtype=torch.float32, device=table.device)
    with _timed('rel_gather[%s,d=%d]' % ('dd.bwd' if backward else 'dd.fwd', d)):
        check(lib().tipk_rel_gather(int(backward), ptr(table), table.stride(0), n, d, rp.n_wg, ptr(rp.wg_rel_ptr),
                                    ptr(rp.unit_meta), ptr(rp.idx), rp.idx_unit, ptr(rp.runs), ptr(rp.node_at),
                                    ptr(row_scale) if backward else None, ptr(out), d, stream_ptr(table.device)),
              'tipk_rel_gather')
    if backward or not reduce:
        return out
    return sum_slabs(out)


def stream_gather_split(n_table, d, max_split=4):
    """column blocks `tipk_stream_gather` would use for a table [n_table, d]; 0 = it does not fit in LDS.
    max_split: 4 on the D-D passes, 16 for the P-P graph (2-column blocks of 8-byte rows)."""
    if switches.on('TIPK_NO_RELLOCAL'):                 # (test hook: the generic fabric-gather route on a small graph)
        return 0
    return int(lib().tipk_stream_gather_supported(n_table, d, max_split))


rel_stream_split = stream_gather_split


def rel_stream_piece():
    return int(lib().tipk_stream_gather_piece())


def dy_products_fused(r, nc, nb):
    """True when `dy_products` takes the fused one-pass kernel for this shape (it alone honours `row_used`)."""
    import ctypes as C
    s_c, s_r = C.c_int(0), C.c_int(0)
    if switches.on('TIPK_NO_DY_FUSED'):
        return False
    check(lib().tipk_rgcn_dy_products_plan(r, nc, nb, C.byref(s_c), C.byref(s_r)), 'tipk_rgcn_dy_products_plan')
    return s_c.value != 0


def stream_gather(sp, table, row_scale=None, write_zeros=True, out=None, label='stream_gather', kind=0, max_split=4,
                  out_scale=None, bias=None, relu=False):
    """out[row] = sum of table rows on a wave-stream plan (include/tipk.h section 1d), table [n_table, d] staged
    in LDS (scaled per row while staged).  write_zeros=False leaves the rows without edges UNTOUCHED: only for
    a consumer that masks them (`dy_products(row_used=...)`) or an `out` buffer that was zeroed once.
    Epilogue of a finished row: relu?(out_scale[row] * sum + bias)."""
    table = _f32c(table)
    require_device(table, sp.ids, row_scale, out_scale, bias)
    d = table.shape[1]
    split = stream_gather_split(sp.n_table, d, max_split)
    assert table.shape[0] == sp.n_table and split and (d // split) * 4 == sp.row_bytes, 'plan was built for another launch shape'
    assert table.stride(1) == 1
    if out is None:
        out = torch.empty((sp.n_rows, d), dtype=torch.float32, device=table.device)
    assert out.shape == (sp.n_rows, d) and out.is_contiguous()
    with _timed('%s[d=%d]' % (label, d)):
        check(lib().tipk_stream_gather(ptr(table), table.stride(0), sp.n_table, d, sp.n_wg, ptr(sp.wave_ptr), ptr(sp.cells),
                                       ptr(sp.ids), sp.idx_unit, ptr(sp.zero_ptr) if write_zeros else None,
                                       ptr(sp.zero_rows), ptr(row_scale), ptr(out), d, kind, max_split, ptr(out_scale),
                                       ptr(bias), int(bool(relu)), stream_ptr(table.device)),
              'tipk_stream_gather')
    return out


def stream_gather_two(sp, table0, table1, out0, out1, label='pair_cells2[dd.fwd]'):
    """out0[row] = sum of table0 rows, out1[row] = sum of table1 rows on ONE wave-stream plan in one launch
    (`tipk_stream_gather_two`): the pair cells of both R-GCN layers of an encoder.  Rows without edges stay untouched."""
    table0, table1 = _f32c(table0), _f32c(table1)
    require_device(table0, table1, sp.ids, out0, out1)
    d = table0.shape[1]
    assert table0.shape == table1.shape == (sp.n_table, d) and table0.stride() == table1.stride() and table0.stride(1) == 1
    assert out0.shape == out1.shape == (sp.n_rows, d) and out0.is_contiguous() and out1.is_contiguous()
    assert stream_gather_split(sp.n_table, d, 1) == 1 and d * 4 == sp.row_bytes, 'plan was built for another launch shape'
    with _timed('%s[d=%d]' % (label, d)):
        check(lib().tipk_stream_gather_two(ptr(table0), ptr(table1), table0.stride(0), sp.n_table, d, sp.n_wg, ptr(sp.wave_ptr),
                                           ptr(sp.cells), ptr(sp.ids), sp.idx_unit, ptr(out0), ptr(out1), d,
                                           stream_ptr(table0.device)), 'tipk_stream_gather_two')


def pair_cells_partner_ok(graph, att, graph2, att2, n):
    """True when the pair cells of a second layer (graph2, att2) can be gathered in the launch that gathers (graph, att)'s:
    the same plan shape (the layers share the D-D graph), att tables of one shape, one column block."""
    if switches.on('TIPK_NO_CELLS_TWO'):
        return False
    p1, p2 = graph.pair_fwd, graph2.pair_fwd
    if p1 is None or p2 is None or att.shape != att2.shape or att.stride() != att2.stride():
        return False
    r, nb = att.shape
    return (p1.n_table == p2.n_table == r and p1.n_rows == p2.n_rows == n * n and p1.n_bands == p2.n_bands and p1.n_wg == p2.n_wg
            and p1.lanes == p2.lanes and p1.idx_unit == p2.idx_unit and p1.symmetric == p2.symmetric
            and stream_gather_split(r, nb, 1) == 1 and nb // 4 == p1.lanes and nb >= 16)


def rel_stream_bwd(sp, table, row_scale=None, write_zeros=True):
    """Transposed D-D pass: table = g [N, d] -> dY (plan of `build_stream_plan`): [R * N, d], or -- compact plans --
    [n_rows + 1, d] node-major with a trailing zero row (a buffer kept ON THE PLAN per width: the zero row is written
    once, every call rewrites all the other rows; it lives exactly as long as the plan, i.e. as the layer's graph, so a
    captured hipGraph that baked its address in never sees it freed or handed to someone else)."""
    if sp.compact is None:
        return stream_gather(sp, table, row_scale, write_zeros, label='rel_stream[dd.bwd]')
    d = table.shape[1]
    key = (d, str(table.device))
    buf = sp.dyc.get(key)
    if buf is None:
        buf = sp.dyc[key] = torch.zeros((sp.n_rows + 1, d), dtype=torch.float32, device=table.device)
    stream_gather(sp, table, row_scale, write_zeros=False, out=buf[:sp.n_rows], label='rel_stream[dd.bwd]')
    return buf


def _strides3(t):
    """(batch stride, row stride, col stride) of a 2-D (batch stride 0) or 3-D tensor."""
    if t.dim() == 2:
        return 0, t.stride(0), t.stride(1)
    return t.stride(0), t.stride(1), t.stride(2)


def _tile_shape(m, n):
    """(BM, BN) the kernel picks (tip_amd/csrc/tipk_gemm.hip: 4x1, 1x4 or 2x2 waves of 32x32)."""
    if n <= 32:
        return 128, 32
    if m <= 32:
        return 32, 128
    return 64, 64


class GemmJob(object):
    """One product prepared for launch: the descriptor, the slab buffer of a split reduction and
    the ordered slab sum that finishes it.  `gemm` runs one job; `gemm_group` runs several
    independent jobs in one grouped launch (+ one grouped slab sum)."""
    __slots__ = ('desc', 'label', 'slabs', 'n_slabs', 'per', 'alpha', 'accumulate', 'out', 'keep', 'gate')


def gemm_job(a, b, out=None, c_in=None, relu=False, alpha=1.0, reduce_batch=False, ksplit=None, kgroup=None):
    """Prepare out = relu?(alpha * a @ b + c_in) on the matrix cores (include/tipk.h section 2).

    a: [M,K] or [Z,M,K]; b: [K,N] or [Z,K,N] -- arbitrary strides (transposed views are free).
    reduce_batch: sum the Z products into one [M,N] (the basis-sum of the R-GCN backward).
    ksplit: K slabs (None = automatic).  Products with few output tiles but a long reduction
    (dW = g^T h over 19 081 proteins, X^T g over 645 drugs, att^T dY over 1 097 relations) would
    run as a handful of serial workgroups; they are cut into slabs that fill the chip and the
    slabs are added in order by `tipk_sum_slabs` (deterministic, no atomics).
    kgroup (with reduce_batch): the Z batch terms are summed in groups of `kgroup` consecutive terms, one
    slab per group -- sum_z a[z] @ b[z] with Z in the hundreds (the pair-form D-D product: one term per
    source node), where one slab per term would be as large as the operands.
    """
    require_device(a, b)
    if a.dtype != torch.float32 or b.dtype != torch.float32:
        raise _lib.TipkError('gemm: fp32 only')
    z = max(a.shape[0] if a.dim() == 3 else 1, b.shape[0] if b.dim() == 3 else 1)
    batched = a.dim() == 3 or b.dim() == 3
    m, k = a.shape[-2], a.shape[-1]
    k2, n = b.shape[-2], b.shape[-1]
    assert k == k2, (a.shape, b.shape)
    a_sz, a_sm, a_sk = _strides3(a)
    b_sz, b_sk, b_sn = _strides3(b)
    reduce_batch = bool(reduce_batch and batched)
    oshape = (z, m, n) if (batched and not reduce_batch) else (m, n)
    dev = a.device
    if out is None:
        out = torch.empty(oshape, dtype=torch.float32, device=dev)
    assert tuple(out.shape) == oshape and out.stride(-1) == 1, (out.shape, oshape, out.stride())
    if c_in is not None:
        assert c_in.shape == out.shape and c_in.stride(-1) == 1

    bm, bn = _tile_shape(m, n)
    tiles = -(-m // bm) * -(-n // bn) * (1 if reduce_batch else z)
    plain = relu is False and (c_in is None or c_in.data_ptr() == out.data_ptr()) and out.is_contiguous()
    slab_mode = None                       # None | 'k' (split the k range) | 'q' (one slab per batch term) | 'g' (per group of terms)
    n_slabs = 0
    if kgroup:
        # (c_in == out: the slab sum accumulates on top of what `out` holds, as in mode 'q')
        assert reduce_batch and plain and z % kgroup == 0 and a.dim() == 3 and b.dim() == 3
        slab_mode, n_slabs = 'g', z // kgroup
    elif reduce_batch and plain and tiles * 2 <= 256 and z > 1:
        slab_mode, n_slabs = 'q', z
    elif ksplit is None:
        # a workgroup's K loop is a chain of dependent ~1.5 us load round trips: small products are cut
        # until a slab is one or two K tiles (the three constants were swept in rounds 2 and 3: all at their optimum)
        grain, wgs = 64, 512
        cap = 320                                               # (K = 19 081: 299 slabs of two K tiles; 128 slabs made five-tile chains)
        # (ceil: K = 645 -> 11 slabs of 64 = two K tiles each; 10 slabs made the chunk 96 = three tiles, three slabs empty)
        want = min(-(-k // grain), -(-wgs // tiles))
        if plain and not reduce_batch and want >= 2 and tiles < 256:
            slab_mode, n_slabs = 'k', int(min(cap, want))
    elif ksplit > 1:
        assert plain and not reduce_batch
        slab_mode, n_slabs = 'k', int(ksplit)

    g = GemmDesc()
    g.m, g.n, g.k = m, n, k
    g.a, g.b = a.data_ptr(), b.data_ptr()
    g.a_sm, g.a_sk, g.b_sk, g.b_sn = a_sm, a_sk, b_sk, b_sn
    g.ksplit = n_slabs if slab_mode == 'k' else 1
    if slab_mode == 'g':
        g.batch, g.kbatch = n_slabs, kgroup
        g.a_sq, g.b_sq, g.a_sz, g.b_sz = a_sz, b_sz, a_sz * kgroup, b_sz * kgroup
    elif reduce_batch and slab_mode != 'q':
        g.batch, g.kbatch = 1, z
        g.a_sq, g.b_sq, g.a_sz, g.b_sz = a_sz, b_sz, 0, 0
    else:
        g.batch, g.kbatch = (z if batched else 1), 1
        g.a_sq, g.b_sq, g.a_sz, g.b_sz = 0, 0, a_sz, b_sz
    slabs = None
    per = m * n * (z if (batched and not reduce_batch) else 1)       # floats of the final result
    if slab_mode == 'k':
        slabs = torch.empty((n_slabs,) + oshape, dtype=torch.float32, device=dev)
        g.c, g.c_sm, g.c_sz, g.c_ss = slabs.data_ptr(), n, m * n, per
    elif slab_mode in ('q', 'g'):
        slabs = torch.empty((n_slabs, m, n), dtype=torch.float32, device=dev)
        g.c, g.c_sm, g.c_sz, g.c_ss = slabs.data_ptr(), n, m * n, 0
    else:
        g.c, g.c_sm = out.data_ptr(), out.stride(-2)
        g.c_sz = out.stride(0) if out.dim() == 3 else 0
        g.c_ss = 0
    if c_in is not None and slab_mode is None:
        g.c_in, g.cin_sm = c_in.data_ptr(), c_in.stride(-2)
        g.cin_sz = c_in.stride(0) if c_in.dim() == 3 else 0
    else:
        g.c_in, g.cin_sm, g.cin_sz = None, 0, 0
    g.alpha, g.relu = (1.0 if slab_mode else alpha), int(relu)

    job = GemmJob()
    job.desc, job.slabs, job.n_slabs, job.per, job.alpha, job.out = g, slabs, n_slabs, per, alpha, out
    job.accumulate = c_in is not None
    job.gate = None                                                  # set by the caller: mask of the finished sum
    job.keep = (a, b, c_in)                                          # operands stay alive until launched
    job.label = '%dx%dx%d,z=%d%s' % (m, n, k, z, ',slabs=%s%d' % (slab_mode, n_slabs) if slab_mode else '')
    return job


def large_kgroup(m, z):
    """Batch-reduced products over a LARGE node set (config 5: [10 000 x 128] x [128 x 128] summed over 32 bases) have few
    output tiles (314) and a reduction of 4 096: four slabs of 8 bases each fill the chip (255 -> 169 us)."""
    return 8 if (m >= 4096 and z >= 16 and z % 8 == 0) else None


def gemm(a, b, out=None, c_in=None, relu=False, alpha=1.0, reduce_batch=False, ksplit=None, kgroup=None):
    """out = relu?(alpha * a @ b + c_in): one product, launched now (see `gemm_job`)."""
    job = gemm_job(a, b, out, c_in, relu, alpha, reduce_batch, ksplit, kgroup)
    st = stream_ptr(a.device)
    with _timed('gemm[%s]' % job.label):
        check(lib().tipk_gemm_f32(job.desc, st), 'tipk_gemm_f32')
        if job.slabs is not None:
            check(lib().tipk_sum_slabs(ptr(job.slabs), job.n_slabs, job.per, job.per, job.alpha,
                                       int(job.accumulate), ptr(job.out), st), 'tipk_sum_slabs')
    return job.out


def dy_products(g_y, att, xb2, row_used=None, n_nodes=0):
    """(d att, d XB) = (g_y @ xb2^T, att^T @ g_y) in ONE pass over g_y [R, N*out] (include/tipk.h
    section 2b); falls back to two grouped GEMMs for shapes the fused kernel does not take."""
    import ctypes as C
    r, nc = g_y.shape
    nb = att.shape[1]
    s_c, s_r = C.c_int(0), C.c_int(0)
    if not switches.on('TIPK_NO_DY_FUSED') and g_y.stride(1) == 1 and att.stride(1) == 1 and xb2.stride(1) == 1:
        check(lib().tipk_rgcn_dy_products_plan(r, nc, nb, C.byref(s_c), C.byref(s_r)), 'tipk_rgcn_dy_products_plan')
    if s_c.value == 0:
        assert row_used is None, 'the two-GEMM path reads every row of dY'
        g_att, g_xb = gemm_group([gemm_job(g_y, xb2.t()), gemm_job(att.t(), g_y)])
        return g_att, g_xb
    dev = g_y.device
    dxb_slabs = torch.empty((s_r.value, nb, nc), dtype=torch.float32, device=dev)
    datt_slabs = torch.empty((s_c.value, r, nb), dtype=torch.float32, device=dev)
    with _timed('dy_products[%dx%dx%d]' % (r, nc, nb)):
        check(lib().tipk_rgcn_dy_products(ptr(g_y), g_y.stride(0), ptr(att), att.stride(0), ptr(xb2), xb2.stride(0),
                                          r, nc, nb, ptr(row_used), n_nodes, ptr(dxb_slabs), ptr(datt_slabs), stream_ptr(dev)),
              'tipk_rgcn_dy_products')
    j_att = slab_job(datt_slabs)
    if s_r.value == 1:                                   # one row range: the slab IS d XB
        gemm_group([], [j_att])
        return j_att.out, dxb_slabs[0]
    j_xb = slab_job(dxb_slabs)
    gemm_group([], [j_xb, j_att])
    return j_att.out, j_xb.out


def node_products_slabs(n_nodes, d, n_rel, nb):
    """d att slabs `node_products` produces for this shape; 0 = shape not supported (dense `dy_products` then)."""
    import ctypes as C
    g = C.c_int(0)
    check(lib().tipk_rgcn_node_products_plan(n_nodes, d, n_rel, nb, C.byref(g)), 'tipk_rgcn_node_products_plan')
    return g.value


def node_products(dyc, cr, att, xb, xbt=None):
    """(pending slab sum of d att, d XB [bases, N, d]) from the COMPACT node-major dY (include/tipk.h section 2d):
    dyc [n_rows + 1, d] as written by `stream_gather` on a compact plan (last row zero), cr = plan.compact,
    xb [bases, N, d]; xbt (optional) the same values as a contiguous [N, d, bases] (read coalesced by the d att product).
    d XB is complete; the d att slabs are summed by the caller's next grouped slab sum."""
    n_rows, d = dyc.shape[0] - 1, dyc.shape[1]
    nb, n = xb.shape[0], xb.shape[1]
    r = att.shape[0]
    assert cr.n_rows == n_rows and xb.shape[2] == d and xb.stride(2) == 1 and att.stride(1) == 1 and dyc.is_contiguous()
    assert xb.stride(0) % 4 == 0 and xb.stride(1) % 4 == 0
    assert xbt is None or (tuple(xbt.shape) == (n, d, nb) and xbt.is_contiguous() and xbt.dtype == torch.float32)
    g = node_products_slabs(n, d, r, nb)
    assert g > 0
    dev = dyc.device
    dxb = torch.empty((nb, n, d), dtype=torch.float32, device=dev)
    datt_slabs = torch.empty((g, r, nb), dtype=torch.float32, device=dev)
    with _timed('node_products[%dx%dx%d,rows=%d]' % (r, n * d, nb, n_rows)):
        check(lib().tipk_rgcn_node_products(ptr(dyc), n_rows, d, ptr(cr.node_desc), ptr(cr.row_rel), ptr(cr.pos),
                                            n, r, ptr(att), att.stride(0), nb, ptr(xb), xb.stride(0),
                                            xb.stride(1), ptr(xbt), ptr(dxb), dxb.stride(0), dxb.stride(1), ptr(datt_slabs),
                                            stream_ptr(dev)), 'tipk_rgcn_node_products')
    return slab_job(datt_slabs), dxb


def dest_products_bits(n_nodes, n_rel, nb, d_in):
    """bits of the relation field of a destination-major edge word (`tipk_rgcn_dest_products`); 0 = shape not supported."""
    if switches.on('TIPK_NO_DEST_FWD'):
        return 0
    return int(lib().tipk_rgcn_dest_products_supported(int(n_nodes), int(n_rel), int(nb), int(d_in)))


def dest_products(dp, x, att):
    """T[b, v, :] = sum over the edges e -> v of att[r_e, b] * x[src_e, :]  (include/tipk.h section 2f) on a `plan.DestPlan`:
    x [N, d_in], att [R, bases] -> T [bases, N, d_in]."""
    x, att = _f32c(x), _f32c(att)
    require_device(x, att, dp.edges)
    n, d_in = x.shape
    r, nb = att.shape
    assert dp.n_nodes == n and dp.n_rel == r
    t = torch.empty((nb, n, d_in), dtype=torch.float32, device=x.device)
    with _timed('dest_products[%dx%dx%d,edges=%d]' % (n, nb, d_in, dp.n_edges)):
        check(lib().tipk_rgcn_dest_products(ptr(x), x.stride(0), d_in, ptr(att), att.stride(0), nb, n, r, ptr(dp.node_desc),
                                            ptr(dp.edges), ptr(t), t.stride(0), t.stride(1), stream_ptr(x.device)),
              'tipk_rgcn_dest_products')
    return t


def row_products_supported(n_nodes, n_rel, nb, channels):
    if switches.on('TIPK_NO_ROW_PRODUCTS'):
        return False
    return bool(lib().tipk_rgcn_row_products_supported(int(n_nodes), int(n_rel), int(nb), int(channels)))


def row_products_s_supported(n_nodes, n_rel, nb, channels):
    if switches.on('TIPK_NO_ROW_PRODUCTS_S'):
        return False
    return bool(lib().tipk_rgcn_row_products_s_supported(int(n_nodes), int(n_rel), int(nb), int(channels)))


def _row_plan(graph, transposed, n, r, nb, channels):
    """The graph's row-stream plan for a pass over rows of `channels` floats, or None.  The kernel's own support query is asked
    BEFORE the (lazily built) plan is touched: building the per-lane plan for a node count it does not take would raise
    (plan.build_row_stream_plan: 16-bit node field), where the pass should fall through to the CSR / gather_sum route."""
    if not graph.has_row_plans:
        return None
    if graph.row_plans_wave_uniform:
        if channels % 64 != 0 or not row_products_s_supported(n, r, nb, channels) or n * channels * 4 >= 2 ** 31:
            return None
    elif not row_products_supported(n, r, nb, channels):
        return None
    return graph.row_bwd if transposed else graph.row_fwd


def row_products(rp, table, att, xb2=None):
    if getattr(rp, 'ROW_BYTES', None) is not None:
        return _row_products_s(rp, table, att, xb2)
    return _row_products_v(rp, table, att, xb2)


def _row_products_s(rp, table, att, xb2=None):
    """`tipk_rgcn_row_products_s` on a `plan.RowStreamPlanS` (wave-uniform entries): same results as the form below."""
    table, att = _f32c(table), _f32c(att)
    require_device(table, att, rp.entries)
    n, ch = table.shape
    r, nb = att.shape
    assert rp.n_nodes == n and rp.n_rel == r and table.stride(1) == 1 and att.stride(1) == 1
    entries = rp.entries_for(table.stride(0) * 4)
    t = torch.empty((nb, n, ch), dtype=torch.float32, device=table.device)
    slabs = None
    if xb2 is not None:
        assert xb2.shape == (nb, n * ch) and xb2.stride(1) == 1
        slabs = torch.empty((-(-n // 8) * (ch // 64), r, nb), dtype=torch.float32, device=table.device)
    with _timed('row_products_s[%dx%dx%d,edges=%d,%s]' % (n, nb, ch, rp.n_edges, 'T+datt' if slabs is not None else 'T')):
        check(lib().tipk_rgcn_row_products_s(ptr(table), table.stride(0), n, ch, ptr(att), att.stride(0), r, nb, ptr(entries),
                                             ptr(rp.desc), ptr(xb2), xb2.stride(0) if xb2 is not None else 0, ptr(t), ptr(slabs),
                                             stream_ptr(table.device)), 'tipk_rgcn_row_products_s')
    if slabs is None:
        return t
    return slab_job(slabs), t


def _row_products_v(rp, table, att, xb2=None):
    """The (relation, node) row sums S = sum of table rows over a row's edges, assembled in LDS, and their products
    (`tipk_rgcn_row_products`, include/tipk.h section 2h) on a `plan.RowStreamPlan`:
    table [N, ch], att [R, bases] -> T [bases, N, ch] = sum_r att[r, b] S[(r, v)];  with xb2 [bases, N * ch] also the slab
    job of d att [R, bases] = sum_v <S[(r, v)], XB[b, v]>  (returns (job, T); the caller's grouped slab sum finishes it)."""
    table, att = _f32c(table), _f32c(att)
    require_device(table, att, rp.entries)
    n, ch = table.shape
    r, nb = att.shape
    assert rp.n_nodes == n and rp.n_rel == r and table.stride(1) == 1 and att.stride(1) == 1
    if table.stride(0) % 64 != 0:                        # (the kernel multiplies `other << 8` by row bytes / 256: rows of 64 k floats)
        wide = torch.empty((n, -(-ch // 64) * 64), dtype=torch.float32, device=table.device)
        wide[:, :ch] = table
        table = wide[:, :ch]
    t = torch.empty((nb, n, ch), dtype=torch.float32, device=table.device)
    slabs = None
    if xb2 is not None:
        assert xb2.shape == (nb, n * ch) and xb2.stride(1) == 1
        slabs = torch.empty((int(lib().tipk_rgcn_row_products_slabs(n, ch)), r, nb), dtype=torch.float32, device=table.device)
    with _timed('row_products[%dx%dx%d,edges=%d,%s]' % (n, nb, ch, rp.n_edges, 'T+datt' if slabs is not None else 'T')):
        check(lib().tipk_rgcn_row_products(ptr(table), table.stride(0), n, ch, ptr(att), att.stride(0), r, nb, ptr(rp.entries),
                                           ptr(rp.desc), ptr(xb2), xb2.stride(0) if xb2 is not None else 0, ptr(t), ptr(slabs),
                                           stream_ptr(table.device)), 'tipk_rgcn_row_products')
    if slabs is None:
        return t
    return slab_job(slabs), t


def sum_slabs_xb(slabs, row_scale, addend, relu, x_out, basis, root, xb_pad):
    """x_out = relu?(row_scale * sum_s slabs[s] + addend) AND, in the same launch, the next layer's row-local products:
    xb_pad[:N, :, :d_out] = x_out basis (node-major, rows padded to 32 columns) and the returned x_out root
    (`tipk_sum_slabs_xb`, include/tipk.h section 2g).  slabs [S, N, 32], basis [bases, 32, d_out], root [32, d_out]."""
    require_device(slabs, x_out, basis, root, xb_pad)
    s_, n, d_in = slabs.shape
    nb, _, d_out = basis.shape
    assert slabs.is_contiguous() and x_out.is_contiguous() and x_out.shape == (n, d_in) and basis.is_contiguous() and root.is_contiguous()
    assert xb_pad.stride()[-2:] == (32, 1) and xb_pad.shape[1] == nb and xb_pad.stride(0) == nb * 32 and xb_pad.shape[0] >= n
    assert addend is None or (addend.is_contiguous() and addend.shape == (n, d_in))
    xroot = torch.empty((n, d_out), dtype=torch.float32, device=slabs.device)
    with _timed('sum_slabs_xb[%dx%d -> %dx%d]' % (s_, n * d_in, nb, d_out)):
        check(lib().tipk_sum_slabs_xb(ptr(slabs), s_, n * d_in, n, d_in, ptr(row_scale), ptr(addend), int(bool(relu)), ptr(x_out),
                                      ptr(basis), ptr(root), nb, d_out, ptr(xb_pad), ptr(xroot), stream_ptr(slabs.device)),
              'tipk_sum_slabs_xb')
    return xroot


def sum_slabs_xb_supported(d_in, d_out):
    return d_in == 32 and 1 <= d_out <= 32 and not switches.on('TIPK_NO_LAYER_HANDOVER')


def _finish_pending(x, pend):
    """Materialise a layer output whose final slab sum was left to its consumer (`_RGCN.forward(defer_output=True)`)."""
    slabs, scale, addend, relu = pend
    sum_slabs(slabs, out=x, row_scale=scale, addend=addend, relu=relu)


def pair_grads_supported(nb, d):
    return bool(lib().tipk_rgcn_pair_grads_supported(int(nb), int(d))) and not switches.on('TIPK_NO_PAIR_BWD')


def pair_grads(pb, cells, xb_pad, g, table=0):
    """The dense half of the pair-form backward pass of an R-GCN layer (include/tipk.h section 2e) on plan `pb`
    (plan.PairBwdPlan): cells = the cell buffer the forward pass filled ([N_pad, N, bases] + its trailing zeros, contiguous),
    xb_pad [N_pad, bases, 32] = the node-major XB buffer of the pair product, g [N, d] (1 / deg is in the plan's slots).
    -> (pg [2 n_alloc + 1, bases]: the gradient row of every linked pair at the place `pair_att_gather` stages it from,
    d XB [bases, N, d] complete)."""
    g = _f32c(g)
    require_device(cells, xb_pad, g, pb.slots)
    n, d = g.shape
    nb = cells.shape[-1]
    assert pb.n_nodes == n and g.stride(1) == 1 and cells.is_contiguous() and xb_pad.stride()[-2:] == (32, 1) and xb_pad.shape[1] == nb
    dev = g.device
    # `table`: which of the plan's gradient tables to fill -- layers that share the plan and whose d att gathers run in ONE
    # launch (tip_amd/encoder.py) need a table each
    key = str(dev) if table == 0 else (str(dev), int(table))
    pg = pb.pg.get(key)
    if pg is None:                                    # kept on the plan, zeroed ONCE: a call rewrites the rows of the linked pairs,
        pg = pb.pg[key] = torch.zeros((2 * pb.n_alloc + 1, nb), dtype=torch.float32, device=dev)   # the others stay zero
    dxb = torch.empty((nb, n, d), dtype=torch.float32, device=dev)
    with _timed('pair_grads[%dx%dx%d,slots=%d]' % (n, nb, d, pb.n_slots)):
        check(lib().tipk_rgcn_pair_grads(ptr(cells), cells.numel() // nb, ptr(xb_pad), ptr(g), g.stride(0), n, nb, d,
                                         ptr(pb.node_desc), ptr(pb.slots), ptr(pb.tile_node), pb.n_slots, ptr(dxb), dxb.stride(0), dxb.stride(1),
                                         ptr(pg), pg.shape[0], stream_ptr(dev)), 'tipk_rgcn_pair_grads')
    return pg, dxb


def pair_att_gather(pb, pg):
    """d att of the pair-form backward pass: per partition of the pairs one slab [R, bases] of sums of (symmetrised) rows of
    pg over the pairs each relation links (`tipk_stream_gather_parts`) -> the pending ordered slab sum (`slab_job`)."""
    require_device(pg, pb.part_first)
    nb = pg.shape[1]
    assert pg.shape[0] == 2 * pb.n_alloc + 1 and pg.is_contiguous()
    gp = pb.gather
    slabs = torch.empty((pb.n_parts, pb.n_rel, nb), dtype=torch.float32, device=pg.device)
    with _timed('pair_att_gather[parts=%d,edges=%d]' % (pb.n_parts, gp.n_edges)):
        check(lib().tipk_stream_gather_parts(ptr(pg), nb, nb, pb.n_alloc, ptr(pb.part_first), pb.part_len, ptr(pb.wg_part), gp.n_wg,
                                             ptr(gp.wave_ptr), ptr(gp.cells), ptr(gp.ids), gp.idx_unit, ptr(gp.zero_ptr),
                                             ptr(gp.zero_rows), ptr(slabs), nb, stream_ptr(pg.device)), 'tipk_stream_gather_parts')
    return slab_job(slabs)


def pair_backward(pb, cells, xb_pad, g):
    """(pending slab sum of d att [R, bases], d XB [bases, N, d]): `pair_grads` + `pair_att_gather`."""
    pg, dxb = pair_grads(pb, cells, xb_pad, g)
    return pair_att_gather(pb, pg), dxb


class SlabJob(object):
    """An ordered slab sum with the fused epilogue of `sum_slabs`, prepared for a grouped launch."""
    __slots__ = ('desc', 'out', 'keep')


def slab_job(slabs, out=None, alpha=1.0, accumulate=False, row_scale=None, addend=None, relu=False, gate=None):
    assert slabs.is_contiguous()
    per = slabs[0].numel()
    if out is None:
        out = torch.empty(slabs.shape[1:], dtype=torch.float32, device=slabs.device)
    assert out.is_contiguous() and out.numel() == per
    if addend is not None:
        assert addend.is_contiguous() and addend.numel() == per
    d = SlabSumDesc()
    d.in_, d.n_slabs, d.slab_stride, d.count = slabs.data_ptr(), slabs.shape[0], per, per
    d.alpha, d.accumulate = alpha, int(accumulate)
    d.row_scale = row_scale.data_ptr() if row_scale is not None else None
    d.cols = slabs.shape[-1]
    d.addend = addend.data_ptr() if addend is not None else None
    d.relu, d.out = int(relu), out.data_ptr()
    if gate is not None:
        assert gate.is_contiguous() and gate.numel() == per
        d.gate = gate.data_ptr()
    job = SlabJob()
    job.desc, job.out, job.keep = d, out, (slabs, row_scale, addend, gate)
    return job


def _slab_desc_of(job):
    d = SlabSumDesc()
    d.in_, d.n_slabs, d.slab_stride, d.count = job.slabs.data_ptr(), job.n_slabs, job.per, job.per
    d.alpha, d.accumulate = job.alpha, int(job.accumulate)
    d.row_scale, d.cols, d.addend, d.relu, d.out = None, 0, None, 0, job.out.data_ptr()
    if job.gate is not None:
        assert job.gate.is_contiguous() and job.gate.numel() == job.per
        d.gate = job.gate.data_ptr()
    return d


def gemm_group(jobs, slab_jobs=(), defer_sums=False):
    """Run independent products in ONE grouped launch, then ONE grouped ordered slab sum for those
    that were split (plus any extra `slab_jobs` that are ready at the same point).  Results are
    bit-identical to running the jobs one by one.  Returns the jobs' outputs.
    defer_sums: do NOT launch the slab sums -- return them as a list of `slab_job`-like riders (the caller hands them
    to the next launch that takes riders: `gather_sum(..., riders=...)`); the jobs' outputs are complete only then."""
    jobs = list(jobs)
    slab_jobs = list(slab_jobs)
    if not jobs and not slab_jobs:
        return []
    dev = (jobs[0].out if jobs else slab_jobs[0].out).device
    st = stream_ptr(dev)
    # two members may share an output only as "written by the GEMM launch" + "slab sum accumulated on top"
    by_out = {}
    for j in jobs:
        by_out.setdefault(j.out.data_ptr(), []).append(j)
    for same in by_out.values():
        if len(same) > 1:
            direct = [j for j in same if j.slabs is None]
            assert len(direct) == 1 and all(j.accumulate for j in same if j.slabs is not None), \
                'group members sharing an output: one un-split product + accumulating slab sums only'
            assert len(same) == 2, 'at most one accumulating member per output (order of the sums)'
    for i in range(0, len(jobs), _lib.GROUP_MAX):
        part = jobs[i:i + _lib.GROUP_MAX]
        arr = (GemmDesc * len(part))(*[j.desc for j in part])
        with _timed('gemm_group[%s]' % ' | '.join(j.label for j in part)):
            check(lib().tipk_gemm_f32_group(arr, len(part), st), 'tipk_gemm_f32_group')
    sums = [_slab_desc_of(j) for j in jobs if j.slabs is not None] + [s.desc for s in slab_jobs]
    if defer_sums:
        riders = []
        for dsc, keep in zip(sums, [j for j in jobs if j.slabs is not None] + list(slab_jobs)):
            r = SlabJob()
            r.desc, r.out, r.keep = dsc, keep.out, keep
            riders.append(r)
        return riders
    for i in range(0, len(sums), _lib.GROUP_MAX):
        part = sums[i:i + _lib.GROUP_MAX]
        arr = (SlabSumDesc * len(part))(*part)
        with _timed('sum_slabs_group[%s]' % ' | '.join('%dx%d' % (d.n_slabs, d.count) for d in part)):
            check(lib().tipk_sum_slabs_group(arr, len(part), st), 'tipk_sum_slabs_group')
    return [j.out for j in jobs]


class WgGemmJob(object):
    """A product whose reduction runs inside one workgroup per output tile (`wg_gemm_group`)."""
    __slots__ = ('desc', 'out', 'keep', 'label')


def wg_gemm_job(a, b, out=None, reduce_batch=False, a2=None, b2=None, gate=None, alpha=1.0):
    """Prepare out = gate?(alpha * (a @ b [summed over the batch] + a2 @ b2)) for `wg_gemm_group` (include/tipk.h
    `tipk_gemm_wg_group`): a [M, K] or [Z, M, K], b [K, N] or [Z, K, N], arbitrary strides; a2 [M, K2], b2 [K2, N]
    optional (only without a surviving batch); gate: a tensor shaped like the output.  Returns None when the shape is
    not taken (reductions beyond 64 K tiles of 32, more than 4096 output tiles): the caller uses `gemm_job` then."""
    require_device(a, b)
    assert a.dtype == torch.float32 and b.dtype == torch.float32
    z = max(a.shape[0] if a.dim() == 3 else 1, b.shape[0] if b.dim() == 3 else 1)
    batched = a.dim() == 3 or b.dim() == 3
    m, k = a.shape[-2], a.shape[-1]
    k_b, n = b.shape[-2], b.shape[-1]
    assert k == k_b, (a.shape, b.shape)
    reduce_batch = bool(reduce_batch and batched)
    oshape = (z, m, n) if (batched and not reduce_batch) else (m, n)
    if out is None:
        out = torch.empty(oshape, dtype=torch.float32, device=a.device)
    assert tuple(out.shape) == oshape and out.stride(-1) == 1, (out.shape, oshape, out.stride())
    a_sz, a_sm, a_sk = _strides3(a)
    b_sz, b_sk, b_sn = _strides3(b)
    d = WgGemmDesc()
    g = d.p
    g.m, g.n, g.k, g.ksplit = m, n, k, 1
    g.a, g.b = a.data_ptr(), b.data_ptr()
    g.a_sm, g.a_sk, g.b_sk, g.b_sn = a_sm, a_sk, b_sk, b_sn
    if reduce_batch:
        g.batch, g.kbatch = 1, z
        g.a_sq, g.b_sq, g.a_sz, g.b_sz = a_sz, b_sz, 0, 0
    else:
        g.batch, g.kbatch = (z if batched else 1), 1
        g.a_sq, g.b_sq, g.a_sz, g.b_sz = 0, 0, a_sz, b_sz
    g.c, g.c_sm = out.data_ptr(), out.stride(-2)
    g.c_sz = out.stride(0) if out.dim() == 3 else 0
    g.c_ss = 0
    g.c_in, g.cin_sm, g.cin_sz = None, 0, 0
    g.alpha, g.relu = alpha, 0
    if a2 is not None:
        require_device(a2, b2)
        assert a2.dim() == 2 and b2.dim() == 2 and a2.shape[0] == m and b2.shape[1] == n and a2.shape[1] == b2.shape[0]
        assert a2.dtype == torch.float32 and b2.dtype == torch.float32 and len(oshape) == 2
        d.a2, d.a2_sm, d.a2_sk = a2.data_ptr(), a2.stride(0), a2.stride(1)
        d.b2, d.b2_sk, d.b2_sn = b2.data_ptr(), b2.stride(0), b2.stride(1)
        d.k2 = a2.shape[1]
    if gate is not None:
        require_device(gate)
        assert tuple(gate.shape) == oshape and gate.stride(-1) == 1 and gate.dtype == torch.float32
        d.gate, d.gate_sm = gate.data_ptr(), gate.stride(-2)
        d.gate_sz = gate.stride(0) if gate.dim() == 3 else 0
    if not lib().tipk_gemm_wg_group_supported(d):
        return None
    job = WgGemmJob()
    job.desc, job.out, job.keep = d, out, (a, b, a2, b2, gate)
    job.label = '%dx%dx%d,z=%d%s' % (m, n, k, z, '+%d' % a2.shape[1] if a2 is not None else '')
    return job


def wg_gemm_group(jobs, slab_jobs=()):
    """Run up to 4 `wg_gemm_job`s and up to 3 ordered slab sums (`slab_job`) in ONE launch; returns the products' outputs."""
    jobs, slab_jobs = list(jobs), list(slab_jobs)
    assert len(jobs) <= _lib.WG_GEMM_MAX and len(slab_jobs) <= _lib.WG_SUMS_MAX and (jobs or slab_jobs)
    dev = (jobs[0].out if jobs else slab_jobs[0].out).device
    arr = (WgGemmDesc * max(1, len(jobs)))(*[j.desc for j in jobs])
    sums = (SlabSumDesc * max(1, len(slab_jobs)))(*[s.desc for s in slab_jobs])
    with _timed('wg_gemm_group[%s%s]' % (' | '.join(j.label for j in jobs),
                                          ''.join(' | sum %dx%d' % (s.desc.n_slabs, s.desc.count) for s in slab_jobs))):
        check(lib().tipk_gemm_wg_group(arr, len(jobs), sums, len(slab_jobs), stream_ptr(dev)), 'tipk_gemm_wg_group')
    return [j.out for j in jobs]


def transpose(x):
    """x^T as a row-major tensor; a view when x is stored column-major already (no launch)."""
    if x.dim() == 2 and x.t().is_contiguous() and x.dtype == torch.float32:
        return x.t()
    x = _f32c(x).contiguous()
    require_device(x)
    out = torch.empty((x.shape[1], x.shape[0]), dtype=torch.float32, device=x.device)
    check(lib().tipk_transpose(ptr(x), x.shape[0], x.shape[1], ptr(out), stream_ptr(x.device)), 'tipk_transpose')
    return out


def rows_affine(x, row_mul=None, row_div=None, gate=None, out=None, accumulate=False):
    """out (+)= x * row_mul / row_div * (gate > 0); x/out/gate may be column-slice views."""
    x = _f32c(x)
    require_device(x)
    rows, cols = x.shape
    if out is None:
        out = torch.empty((rows, cols), dtype=torch.float32, device=x.device)
        assert not accumulate
    assert out.shape == x.shape and out.stride(1) == 1
    if gate is not None:
        gate = _f32c(gate)
    check(lib().tipk_rows_affine(ptr(x), x.stride(0), ptr(row_mul), ptr(row_div), ptr(gate),
                                 gate.stride(0) if gate is not None else 0, ptr(out), out.stride(0), rows, cols,
                                 int(accumulate), stream_ptr(x.device)), 'tipk_rows_affine')
    return out


def gate_colsum(x, gate):
    """(x * (gate > 0), partial column sums [G, 1, cols] of it) in one pass; None if unsupported.
    (Round 4: letting the launch's LAST workgroup add the partial rows behind a ticket -- no sum_slabs launch -- made the
    step 16 us SLOWER: handing data between workgroups inside a launch takes a device-scope release in every workgroup,
    i.e. an L2 write-back per XCD on MI355X.)"""
    x, gate = _f32c(x), _f32c(gate)
    require_device(x, gate)
    rows, cols = x.shape
    groups = int(lib().tipk_gate_colsum_groups(rows, cols))
    if groups == 0:
        return None
    out = torch.empty((rows, cols), dtype=torch.float32, device=x.device)
    scratch = torch.empty((groups, 1, cols), dtype=torch.float32, device=x.device)
    check(lib().tipk_gate_colsum(ptr(x), x.stride(0), ptr(gate), gate.stride(0), ptr(out), out.stride(0), rows, cols,
                                 ptr(scratch), stream_ptr(x.device)), 'tipk_gate_colsum')
    return out, scratch


def col_sum(x):
    x = _f32c(x)
    require_device(x)
    rows, cols = x.shape
    scratch = torch.empty((256, cols), dtype=torch.float32, device=x.device)
    out = torch.empty((cols,), dtype=torch.float32, device=x.device)
    check(lib().tipk_col_sum(ptr(x), x.stride(0), rows, cols, ptr(scratch), ptr(out), stream_ptr(x.device)),
          'tipk_col_sum')
    return out


def _idx_bytes(t):
    if t.dtype == torch.int64:
        return 8
    if t.dtype == torch.int32:
        return 4
    raise _lib.TipkError('index tensors must be int32 or int64, got %s' % t.dtype)


def _uv(edge_index):
    ei = edge_index if edge_index.is_contiguous() else edge_index.contiguous()
    return ei[0], ei[1]


# Derived facts about an index tensor (range verdict, packed copy) are kept in a SIDE TABLE keyed by the tensor object,
# not as attributes on it: `torch.save(model)` (tip.py:36) pickles a tensor's __dict__, so attributes would put a 4 E-byte
# packed copy into every checkpoint and restore a stale "range ok" verdict on load (a loaded tensor's version counter
# restarts at 0).  An entry dies with its tensor (weakref finalizer) and is only trusted while the tensor's version,
# storage address and shape are what they were when it was made.
_TENSOR_FACTS = {}


def _facts(t):
    """The side-table entry (a dict) of tensor `t`, emptied if `t` was modified, re-pointed or re-shaped since."""
    import weakref
    tag = (t._version, t.data_ptr(), tuple(t.shape), str(t.device))
    ent = _TENSOR_FACTS.get(id(t))
    if ent is None or ent[0]() is not t:
        ent = _TENSOR_FACTS[id(t)] = [weakref.ref(t), tag, {}]
        weakref.finalize(t, _TENSOR_FACTS.pop, id(t), None)
    elif ent[1] != tag:
        ent[1], ent[2] = tag, {}
    return ent[2]


def _range_checked(t, limit, what):
    """Raise IndexError unless every entry of the index tensor `t` lies in [0, limit).  The verdict is remembered per
    tensor OBJECT (`_facts`: keyed by version counter, storage address, shape): no global cache pins index tensors, a
    tensor that was modified in place -- or unpickled -- is checked again, and a fresh tensor costs ONE fused device
    reduction and one sync."""
    facts = _facts(t)
    if facts.get('range_ok') == int(limit):
        return
    if t.numel():
        lo, hi = torch.stack([t.min(), t.max()]).tolist()
        if lo < 0 or hi >= limit:
            raise IndexError('%s id out of range: [%d, %d] for %d' % (what, lo, hi, limit))
    facts['range_ok'] = int(limit)


def validate_triples(edge_index, edge_type, n_nodes, n_rel):
    """Range check of a triple list, done ONCE per tensor (the verdict is stored on the tensor, so the per-step cost
    is an attribute lookup): the decoder kernels index the LDS images of z / d z with 16-bit node ids and `weight`
    rows with the relation id, so an out-of-range id would corrupt memory silently where the reference raises
    IndexError.  (A first-time check synchronises: do not meet it inside a hipGraph capture -- the warm-up steps of
    tip_amd.train.GraphedTrainStep take care of that.)"""
    n = edge_index.shape[-1]
    if n >= 2 ** 31:
        raise ValueError('triple lists are indexed with 32-bit positions: %d >= 2^31' % n)
    _range_checked(edge_index, n_nodes, 'node')
    if edge_type is not None:
        if edge_type.numel() != n:
            raise ValueError('edge_type has %d entries for %d triples' % (edge_type.numel(), n))
        _range_checked(edge_type, n_rel, 'relation')


_TASK_CACHE = {}
TASK_POSITIONS = 2048          # = TASK_MAX in tipk_distmult.hip (a task's ids are staged in LDS)


def relation_tasks(edge_type, pos_index=None):
    """int32 [T,4] (relation, begin, end, pos_weight) tasks for the LDS-resident decoder kernels, or
    None when the triples are not grouped by relation (include/tipk.h section 4).  Built once per tensor.

    pos_index: the positive triples of the fused objective.  If every relation's block is
    [pairs | the same pairs mirrored] (the data contract of the path, src/utils.py:35-65 -- verified
    here, element by element), the first half gets weight 2 and the mirrored half weight 0."""
    key = (edge_type.data_ptr(), tuple(edge_type.shape), edge_type._version, str(edge_type.device),
           None if pos_index is None else (pos_index.data_ptr(), pos_index._version))
    hit = _TASK_CACHE.get(key)
    if hit is not None:
        return hit[0]
    tasks = None
    if edge_type.numel():
        rels, counts = torch.unique_consecutive(edge_type, return_counts=True)
        if torch.unique(rels).numel() == rels.numel():                      # each relation is one run
            start = torch.cumsum(counts, 0) - counts
            dev = rels.device
            mirrored = False
            if pos_index is not None and not switches.on('TIPK_NO_SYMMETRIC_POS') and bool((counts % 2 == 0).all()):
                half = counts // 2
                pos = torch.arange(edge_type.numel(), device=dev)
                run_of = torch.repeat_interleave(torch.arange(rels.numel(), device=dev), counts)
                first = pos < (start + half)[run_of]
                partner = torch.where(first, pos + half[run_of], pos - half[run_of])
                u, v = pos_index[0], pos_index[1]
                mirrored = bool(((u == v[partner]) & (v == u[partner])).all())
            if mirrored:                                                    # two segments per relation
                seg_rel = torch.repeat_interleave(rels, 2)
                seg_cnt = torch.repeat_interleave(counts // 2, 2)
                seg_start = torch.stack([start, start + counts // 2], dim=1).view(-1)
                seg_w = torch.tensor([2, 0], device=dev).repeat(rels.numel())
            else:
                seg_rel, seg_cnt, seg_start = rels, counts, start
                seg_w = torch.ones_like(rels)
            n_chunks = (seg_cnt + TASK_POSITIONS - 1) // TASK_POSITIONS
            run = torch.repeat_interleave(torch.arange(seg_rel.numel(), device=dev), n_chunks)
            first_c = torch.cumsum(n_chunks, 0) - n_chunks
            local = torch.arange(run.numel(), device=dev) - first_c[run]
            begin = seg_start[run] + local * TASK_POSITIONS
            end = torch.minimum(begin + TASK_POSITIONS, seg_start[run] + seg_cnt[run])
            # heavier tasks first: a weight-2 / weight-1 position costs two evaluations, a weight-0 one
            cost = (end - begin) * (1 + (seg_w[run] > 0).long())
            order = torch.sort(cost, descending=True, stable=True).indices
            tasks = torch.stack([seg_rel[run], begin, end, seg_w[run]], dim=1)[order].to(torch.int32).contiguous()
    if len(_TASK_CACHE) > 16:
        _TASK_CACHE.clear()
    _TASK_CACHE[key] = (tasks, edge_type, pos_index)                         # pin the tensors: pointers stay unique
    return tasks


def distmult_fwd(z, weight, edge_index, edge_type, sigmoid=True):
    z, weight = _f32c(z).contiguous(), _f32c(weight).contiguous()
    require_device(z, weight, edge_index, edge_type)
    validate_triples(edge_index, edge_type, z.shape[0], weight.shape[0])
    u, v = _uv(edge_index)
    et = edge_type.contiguous()
    n = u.numel()
    score = torch.empty((n,), dtype=torch.float32, device=z.device)
    check(lib().tipk_distmult_fwd(ptr(z), z.shape[0], z.shape[1], ptr(weight), weight.shape[0], ptr(u), ptr(v),
                                  _idx_bytes(u), ptr(et), _idx_bytes(et), n, int(sigmoid), ptr(score),
                                  stream_ptr(z.device)), 'tipk_distmult_fwd')
    return score


def distmult_bwd(g_score, score, z, weight, edge_index, edge_type, sigmoid=True):
    z, weight = _f32c(z).contiguous(), _f32c(weight).contiguous()
    g_score = _f32c(g_score).contiguous()
    validate_triples(edge_index, edge_type, z.shape[0], weight.shape[0])
    u, v = _uv(edge_index)
    et = edge_type.contiguous()
    g_z, g_w = torch.zeros_like(z), torch.zeros_like(weight)
    tasks = relation_tasks(et)
    check(lib().tipk_distmult_bwd(ptr(g_score), ptr(score), ptr(z), z.shape[0], z.shape[1], ptr(weight),
                                  weight.shape[0], ptr(u), ptr(v), _idx_bytes(u), ptr(et), _idx_bytes(et),
                                  u.numel(), int(sigmoid), ptr(tasks), 0 if tasks is None else tasks.shape[0],
                                  ptr(g_z), ptr(g_w), stream_ptr(z.device)),
          'tipk_distmult_bwd')
    return g_z, g_w


_DET_WS = {}


def _det_workspace(device, n_nodes, k, n_rel):
    """Zeroed fixed-point workspace of the deterministic objective (one per device and shape; every
    call leaves it zeroed again, so it is allocated and cleared exactly once)."""
    key = (str(device), int(n_nodes), int(k), int(n_rel))
    ws = _DET_WS.get(key)
    if ws is None:
        n = int(lib().tipk_distmult_workspace_bytes(n_nodes, k, n_rel)) // 8
        ws = _DET_WS[key] = torch.zeros(n, dtype=torch.int64, device=device)
    return ws


def packed_pairs(edge_index, n_nodes):
    """int32 [E]: the pair (u, v) of every triple as ONE 32-bit word u | v << 16 (include/tipk.h section 4, idx_bytes = 2).
    Built once per tensor version (the positives of the path are static) and kept in the side table (`_facts`: it is
    derived data and must not travel with `torch.save(model)`)."""
    assert n_nodes <= 65535
    facts = _facts(edge_index)
    w = facts.get('packed')
    if w is None:
        w = (edge_index[0].to(torch.int64) | (edge_index[1].to(torch.int64) << 16))
        w = facts['packed'] = torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32).contiguous()
    return w


def unpack_pairs(packed):
    """int64 [2, E] from packed pairs (for callers that need the reference's form)."""
    w = packed.to(torch.int64) & 0xffffffff
    out = torch.stack([w & 0xffff, w >> 16])
    out._tipk_sampled = getattr(packed, '_tipk_sampled', False)
    return out


def distmult_loss(z, weight, pos_index, neg_index, edge_type, need_grad=True):
    """(loss [1], g_z, g_w) of the fused TIP objective (include/tipk.h section 4); bitwise reproducible
    (fixed-point cross-workgroup sums) unless TIPK_FLOAT_ATOMICS=1.
    neg_index: int64 / int32 [2, E] like pos_index, or the sampler's PACKED form (int32 [E], u | v << 16:
    `typed_negative_sampling(..., packed=True)`) -- the positives are then narrowed once to the same form and the kernel
    reads 8 bytes of ids per position instead of 32."""
    z, weight = _f32c(z).contiguous(), _f32c(weight).contiguous()
    require_device(z, weight, pos_index, neg_index, edge_type)
    validate_triples(pos_index, edge_type, z.shape[0], weight.shape[0])
    packed = bool(getattr(neg_index, '_tipk_packed_pairs', False))
    et = edge_type.contiguous()
    if packed:
        tasks = relation_tasks(et, pos_index)
        if (tasks is None or z.shape[1] not in (4, 8, 16) or switches.on('TIPK_FLOAT_ATOMICS') or z.shape[0] > 65535
                or neg_index.numel() != pos_index.shape[1]):
            neg_index, packed = unpack_pairs(neg_index).type_as(pos_index), False      # the general kernels take plain ids
    if packed and not getattr(neg_index, '_tipk_sampled', False):      # a packed tensor that is not the sampler's own output
        validate_triples(unpack_pairs(neg_index), None, z.shape[0], weight.shape[0])
    if packed:                                             # (`_store`: the finalize launch overwrites -- no zero fills)
        loss = torch.empty((1,), dtype=torch.float32, device=z.device)
        g_z = torch.empty_like(z) if need_grad else None
        g_w = torch.empty_like(weight) if need_grad else None
        pp = packed_pairs(pos_index, z.shape[0])
        ws = _det_workspace(z.device, z.shape[0], z.shape[1], weight.shape[0])
        st = lib().tipk_distmult_loss_store(ptr(z), z.shape[0], z.shape[1], ptr(weight), weight.shape[0], ptr(pp), None,
                                            ptr(neg_index), None, 2, ptr(et), _idx_bytes(et), pp.numel(),
                                            ptr(tasks), tasks.shape[0], ptr(loss), ptr(g_z), ptr(g_w), ptr(ws),
                                            stream_ptr(z.device))
        if st != -2:                                       # TIPK_EUNSUPPORTED: z does not fit the kernel's LDS image
            check(st, 'tipk_distmult_loss_store')
            return loss, g_z, g_w
        neg_index = unpack_pairs(neg_index).type_as(pos_index)
    if not getattr(neg_index, '_tipk_sampled', False):
        validate_triples(neg_index, None, z.shape[0], weight.shape[0])       # sampler output is in range by construction
    pu, pv = _uv(pos_index)
    nu, nv = _uv(neg_index)
    assert pu.dtype == nu.dtype and pu.numel() == nu.numel()
    tasks = relation_tasks(et, pos_index)
    ws = None if (tasks is None or switches.on('TIPK_FLOAT_ATOMICS')) else \
        _det_workspace(z.device, z.shape[0], z.shape[1], weight.shape[0])
    if ws is not None:                                     # deterministic path: its finalize launch overwrites the outputs
        loss = torch.empty((1,), dtype=torch.float32, device=z.device)
        g_z = torch.empty_like(z) if need_grad else None
        g_w = torch.empty_like(weight) if need_grad else None
        st = lib().tipk_distmult_loss_store(ptr(z), z.shape[0], z.shape[1], ptr(weight), weight.shape[0], ptr(pu), ptr(pv),
                                            ptr(nu), ptr(nv), _idx_bytes(pu), ptr(et), _idx_bytes(et), pu.numel(),
                                            ptr(tasks), tasks.shape[0], ptr(loss), ptr(g_z), ptr(g_w), ptr(ws),
                                            stream_ptr(z.device))
        if st != -2:
            check(st, 'tipk_distmult_loss_store')
            return loss, g_z, g_w
    loss = torch.zeros((1,), dtype=torch.float32, device=z.device)
    g_z = torch.zeros_like(z) if need_grad else None
    g_w = torch.zeros_like(weight) if need_grad else None
    check(lib().tipk_distmult_loss(ptr(z), z.shape[0], z.shape[1], ptr(weight), weight.shape[0], ptr(pu), ptr(pv),
                                   ptr(nu), ptr(nv), _idx_bytes(pu), ptr(et), _idx_bytes(et), pu.numel(),
                                   ptr(tasks), 0 if tasks is None else tasks.shape[0],
                                   ptr(loss), ptr(g_z), ptr(g_w), ptr(ws), stream_ptr(z.device)), 'tipk_distmult_loss')
    return loss, g_z, g_w


def typed_negative_sampling_device(pos_key_sorted, rel_ptr, n_rel, n_nodes, seed, n_positions, dtype=torch.int64,
                                   call_counter=None, wg=None, pos_offset=None, packed=False, keys32=None):
    """pos_offset: optional int64 device tensor [n_rel]: Philox counter of position e of relation r = e + pos_offset[r]
    (relation-sharded runs: the position's number in the whole triple list).
    call_counter: optional int64 device tensor [2] = {position, seed} (the stream's state): the
    Philox key is derived on the device from it, `seed` is ignored, and the position is advanced by
    one afterwards."""
    require_device(pos_key_sorted, rel_ptr, call_counter)
    dev = pos_key_sorted.device
    st = stream_ptr(dev)
    wg_ptr, wg_units = wg if (wg is not None and not switches.on('TIPK_NO_BITMAP')) else (None, None)
    # a stream state with a ticket word {position, seed, ticket}: the sampling launch moves the position on itself
    adv = 1 if (call_counter is not None and call_counter.numel() >= 3 and n_positions > 0 and n_rel > 0) else 0
    if packed:                                     # one 32-bit word u | v << 16 per position (same draws, same pairs)
        assert n_nodes <= 65535
        out = torch.empty((n_positions,), dtype=torch.int32, device=dev)
        check(lib().tipk_typed_negative_sampling(ptr(pos_key_sorted), ptr(rel_ptr), n_rel, n_nodes, seed, ptr(call_counter), adv,
                                                 ptr(wg_ptr), ptr(wg_units), 0 if wg_ptr is None else wg_ptr.numel() - 1,
                                                 ptr(pos_offset), ptr(keys32), ptr(out), None, 2, n_positions, st),
              'tipk_typed_negative_sampling')
        out._tipk_packed_pairs = True
    else:
        out = torch.empty((2, n_positions), dtype=dtype, device=dev)
        check(lib().tipk_typed_negative_sampling(ptr(pos_key_sorted), ptr(rel_ptr), n_rel, n_nodes, seed,
                                                 ptr(call_counter), adv, ptr(wg_ptr), ptr(wg_units),
                                                 0 if wg_ptr is None else wg_ptr.numel() - 1, ptr(pos_offset), ptr(keys32), ptr(out[0]), ptr(out[1]),
                                                 8 if dtype == torch.int64 else 4, n_positions, st),
              'tipk_typed_negative_sampling')
    if call_counter is not None and not adv:
        check(lib().tipk_counter_advance(ptr(call_counter), st), 'tipk_counter_advance')
    out._tipk_sampled = True                       # ids < n_nodes by construction: no range check downstream
    return out


# ---------------------------------------------------------------------------------------------
# static graph containers (plans in both directions + normalisers)
# ---------------------------------------------------------------------------------------------
PAIR_FWD_MAX_WORLD = 4   # fallback rule when the forward routes cannot be timed (first call under capture): >= 4 ranks -> Y route
PAIR_KGROUP = 8          # source nodes whose products are summed inside one wavefront of the pair-form product


def pair_product(cells, xb_nb, symmetric=False, links=None, zeros=None, xbt=None):
    """slabs[g] = sum_{u in group g} cells[u] (N x bases) @ xb_nb[u] (bases x out)  (include/tipk.h section 2c):
    cells [N_pad, N, bases], xb_nb [N_pad, bases, out] -- or a column slice [..., :out] of a buffer whose rows are
    padded to 32 columns with zeros (what the dedicated kernel reads) -> [N_pad / PAIR_KGROUP, N, out].
    links (optional, uint32 [N_pad, ceil(N / 32)] as int32): bit r of word (u, t) = pair (u, 32 t + r) is linked; the cells of the
    other pairs are not read (they are zero).
    xbt (optional, [N, out, bases] contiguous): filled with XB of the N nodes, bases innermost (`node_products` reads it)."""
    n_pad, n, nb = cells.shape
    d = xb_nb.shape[2]
    assert xbt is None or (tuple(xbt.shape) == (n, d, nb) and xbt.is_contiguous() and xbt.dtype == torch.float32)
    assert cells.is_contiguous() and xb_nb.shape[:2] == (n_pad, nb) and n_pad % PAIR_KGROUP == 0
    padded = xb_nb.stride() == (nb * 32, 32, 1) and d <= 32
    if (not lib().tipk_pair_product_supported(nb, d) or switches.on('TIPK_NO_PAIR_PRODUCT') or not padded) and not symmetric:
        job = gemm_job(cells, xb_nb, reduce_batch=True, kgroup=PAIR_KGROUP)      # same sums on the tiled GEMM
        with _timed('gemm[%s]' % job.label):
            check(lib().tipk_gemm_f32(job.desc, stream_ptr(cells.device)), 'tipk_gemm_f32')
        if xbt is not None:
            xbt.copy_(xb_nb[:n].permute(0, 2, 1))
        return job.slabs
    assert padded, 'tipk_pair_product reads XB rows padded to 32 columns (AggGraph.pair_buffers)'
    slabs = torch.empty((n_pad // PAIR_KGROUP, n, d), dtype=torch.float32, device=cells.device)
    with _timed('pair_product[%dx%dx%dx%d]' % (n_pad, n, nb, d)):
        check(lib().tipk_pair_product(ptr(cells), ptr(xb_nb), n_pad, n, nb, d, PAIR_KGROUP, int(symmetric), ptr(links), ptr(zeros),
                                      ptr(xbt), ptr(slabs), stream_ptr(cells.device)), 'tipk_pair_product')
    return slabs


class AggGraph(object):
    """fwd: out rows <- table rows;  bwd: the transpose.  scale = per-out-row factor (1/deg)."""

    def __init__(self, fwd, bwd, scale=None, rl_fwd=None, rl_bwd=None, bwd_scaled=False, csr_bwd=None, rs_bwd=None,
                 pair_fwd=None, pair_bwd=None, dest_fwd=None, row_fwd=None, row_bwd=None, row_wave_uniform=False):
        """fwd / bwd: GatherPlans, or zero-argument callables that build them on first use (the
        generic D-D plans are only needed where the relation-local kernel does not apply).
        csr_bwd: optional callable -> CsrPlan of the transposed pass (every row written, rows short)."""
        self._fwd, self._bwd, self.scale = fwd, bwd, scale
        self._csr_bwd = csr_bwd
        self._rl_fwd, self._rl_bwd = rl_fwd, rl_bwd        # relation-local (LDS) plans of a D-D graph: plans, or callables
                                                           # that build them on first use (fallback routes only)
        self._rs_bwd = rs_bwd                              # wave-stream plan of the transposed pass (LDS-resident g'), or a
                                                           # callable that builds it on first use (the pair-form backward
                                                           # pass never needs it)
        self.pair_fwd = pair_fwd                           # wave-stream plan of the forward pass in pair form (LDS-resident att)
        self._pair_bwd = pair_bwd                          # plan.PairBwdPlan of the pair-form backward pass (or a callable)
        self._dest_fwd = dest_fwd                          # plan.DestPlan of the forward pass of large graphs (or a callable)
        self._row_fwd, self._row_bwd = row_fwd, row_bwd    # plan.RowStreamPlans of large graphs (rows by destination / by
                                                           # source), or callables: tipk_rgcn_row_products
        self.has_row_plans = row_fwd is not None and row_bwd is not None
        self.row_plans_wave_uniform = bool(row_wave_uniform)   # plan.RowStreamPlanS (widths % 64 == 0) | plan.RowStreamPlan
        self.pair_stamp = 0                                # bumped by every pass that rewrites the pair buffers
        self._pair_cells = {}                              # persistent cell / XB buffers of the pair form, zeroed once
        self.bwd_scaled = bwd_scaled                       # bwd plan's edge weights already carry `scale`
        self.fwd_route = {}                                # sharded layers: timed choice pair form | Y route (ops._fwd_route)

    @property
    def rs_bwd(self):
        if callable(self._rs_bwd):
            self._rs_bwd = self._rs_bwd()
        return self._rs_bwd

    @property
    def row_fwd(self):
        if callable(self._row_fwd):
            self._row_fwd = self._row_fwd()
        return self._row_fwd

    @property
    def row_bwd(self):
        if callable(self._row_bwd):
            self._row_bwd = self._row_bwd()
        return self._row_bwd

    @property
    def dest_fwd(self):
        if callable(self._dest_fwd):
            self._dest_fwd = self._dest_fwd()
        return self._dest_fwd

    @property
    def pair_bwd(self):
        if callable(self._pair_bwd):
            self._pair_bwd = self._pair_bwd()
        return self._pair_bwd

    @property
    def rl_fwd(self):
        if callable(self._rl_fwd):
            self._rl_fwd = self._rl_fwd()
        return self._rl_fwd

    @property
    def rl_bwd(self):
        if callable(self._rl_bwd):
            self._rl_bwd = self._rl_bwd()
        return self._rl_bwd

    @property
    def fwd(self):
        if callable(self._fwd):
            self._fwd = self._fwd()
        return self._fwd

    @property
    def bwd(self):
        if callable(self._bwd):
            self._bwd = self._bwd()
        return self._bwd

    def pair_buffers(self, n, nb, d_out, device):
        """(cells C[u, v, b] as [N_pad, N, bases], XB as [N_pad, bases, out] -- a column slice of rows padded to 32)
        of the pair-form forward, N_pad = N
        rounded up to PAIR_KGROUP.  Allocated and zeroed ONCE: every call rewrites the cells of the linked drug
        pairs and the first N blocks of XB; nothing ever touches the rest."""
        key = (int(n), int(nb), int(d_out), str(device))
        buf = self._pair_cells.get(key)
        if buf is None:
            n_pad = -(-n // PAIR_KGROUP) * PAIR_KGROUP
            xb_pad = torch.zeros((n_pad, nb, 32 if d_out <= 32 else d_out), dtype=torch.float32, device=device)
            flat = torch.zeros(n_pad * n * nb + 64, dtype=torch.float32, device=device)     # (+ a block of zeros behind the cells)
            buf = self._pair_cells[key] = (flat[:n_pad * n * nb].view(n_pad, n, nb),
                                           xb_pad[:, :, :d_out],           # rows padded to 32 columns: the product kernel's layout
                                           flat[n_pad * n * nb:])
        return buf

    def xbt_buffer(self, n, nb, d_out, device):
        """XB a second time as [N, out, bases] -- what the d att product of the backward pass reads (`node_products`): a
        column of all bases is one 128-byte line there (it is 32 lines in the node-major buffer of the pair product)."""
        key = ('xbt', int(n), int(nb), int(d_out), str(device))
        buf = self._pair_cells.get(key)
        if buf is None:
            buf = self._pair_cells[key] = torch.empty((n, d_out, nb), dtype=torch.float32, device=device)
        return buf

    @property
    def csr_bwd(self):
        if callable(self._csr_bwd):
            self._csr_bwd = self._csr_bwd()
        return self._csr_bwd


# ---------------------------------------------------------------------------------------------
# autograd Functions
# ---------------------------------------------------------------------------------------------
class _Aggregate(torch.autograd.Function):
    """out = scale * (A table) + bias, optional fused ReLU; grad: A^T (scale * g)."""

    @staticmethod
    def forward(ctx, table, bias, graph, relu):
        out = gather_sum(graph.fwd, table, row_scale=graph.scale, bias=bias, relu=relu)
        ctx.graph, ctx.relu, ctx.has_bias = graph, relu, bias is not None
        ctx.save_for_backward(out if relu else None)
        return out

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        graph = ctx.graph
        g = _f32c(g)
        # pre = scale * agg + bias, out = relu(pre):  g_pre = g (.) [out > 0];  g_agg = scale * g_pre
        g_pre = rows_affine(g, gate=out) if ctx.relu else g
        g_bias = None
        if ctx.has_bias:
            g_bias = col_sum(g_pre)
        g_agg = g_pre
        if graph.scale is not None and not graph.bwd_scaled:
            g_agg = rows_affine(g_pre, row_mul=graph.scale)
        g_table = gather_sum(graph.bwd, g_agg) if ctx.needs_input_grad[0] else None
        return g_table, g_bias, None, None


def aggregate(table, graph, bias=None, relu=False):
    return _Aggregate.apply(table, bias, graph, relu)


class _Linear(torch.autograd.Function):
    """y = x @ W^T with W [out, in] (the `lin` of GCNConv); x=None means identity features."""

    @staticmethod
    def forward(ctx, x, weight):
        ctx.identity = x is None
        if x is None:
            ctx.save_for_backward(weight)
            return transpose(weight)
        x = _f32c(x)
        ctx.save_for_backward(x, weight)
        return gemm(x, weight.t())

    @staticmethod
    def backward(ctx, g):
        g = _f32c(g)
        if ctx.identity:
            (weight,) = ctx.saved_tensors
            return None, (g.contiguous().t() if weight.t().is_contiguous() else transpose(g))
        x, weight = ctx.saved_tensors
        jobs = [gemm_job(g.t(), x)]
        if ctx.needs_input_grad[0]:
            jobs.append(gemm_job(g, weight))
        outs = gemm_group(jobs)                                          # d W and d x: one grouped launch
        return (outs[1] if len(outs) > 1 else None), outs[0]


def linear_t(x, weight):
    return _Linear.apply(x, weight)


class _MatMul(torch.autograd.Function):
    """y = x @ w  (plain dense product on the matrix cores)."""

    @staticmethod
    def forward(ctx, x, w):
        x, w = _f32c(x), _f32c(w)
        ctx.save_for_backward(x, w)
        return gemm(x, w)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = _f32c(g)
        j_w = gemm_job(x.t(), g) if ctx.needs_input_grad[1] else None
        j_x = gemm_job(g, w.t()) if ctx.needs_input_grad[0] else None
        gemm_group([j for j in (j_w, j_x) if j is not None])
        return (j_x.out if j_x else None), (j_w.out if j_w else None)


def matmul(x, w):
    return _MatMul.apply(x, w)


FWD_ROUTE_MODE = os.environ.get('TIPK_FWD_ROUTE', 'rule')     # 'rule' | 'pair' | 'y' | 'timed' (bench.py sets 'timed' for sharded runs)


def _fwd_route(graph, x, basis, att, pair, shard):
    """'pair' | 'y': the forward route of a relation-SHARDED layer.  The two routes give the same partial aggregate up to
    ROUNDING, so the choice must not depend on anything that varies from run to run or from rank to rank:

      'rule'   (default) the rule of thumb from the 1-GPU timings: PAIR_FWD_MAX_WORLD ranks or more -> Y route;
      'pair' / 'y'       pinned (parity runs);
      'timed'  both routes are timed once per (graph, layer width) on this rank's relations -- HIP events on the launch
               stream, 3 runs each after a warm-up -- the times are SUMMED OVER THE RANKS (one all-reduce of two floats) and
               every rank takes the route with the smaller sum: one decision for the whole job, recorded on the graph
               (bench.py prints it as `config.forward_routes`).  A first call under graph capture cannot time anything and
               uses the rule.

    EVERY rank of a sharded layer calls this, with pair = None when it cannot take the pair form itself (no relations, an
    att table beyond the LDS, a plan of another shape): in 'timed' mode the all-reduce below is part of the layer's
    collective sequence, so whether it happens must not depend on a rank's own shard -- such a rank adds +inf to the pair
    form's sum (the job takes the Y route) and the time of its own Y route (0 without relations)."""
    n = x.shape[0]
    nb, _, d_out = basis.shape
    r = att.shape[0]
    key = (int(n), int(nb), int(d_out))
    hit = graph.fwd_route.get(key)
    if hit is not None:
        return hit[0] if pair is not None else 'y'
    mode = FWD_ROUTE_MODE
    if mode in ('pair', 'y'):
        return mode if pair is not None else 'y'
    rule = 'y' if shard.world >= PAIR_FWD_MAX_WORLD else 'pair'
    if mode != 'timed' or torch.cuda.is_current_stream_capturing() or _TIMING is not None:
        return rule if pair is not None else 'y'
    with torch.no_grad():
        routes = []
        if r > 0:
            xb = gemm(x, basis)
            use_rl = rel_gather_usable(graph.rl_fwd, n, d_out, False)

            def y_route():
                y = gemm(att, xb.view(nb, n * d_out)).view(r * n, d_out)
                return sum_slabs(rel_gather(graph.rl_fwd, y, backward=False, reduce=False)) if use_rl else gather_sum(graph.fwd, y)
            routes.append(('y', y_route))
        if pair is not None:
            cells, xb_nb, _ = graph.pair_buffers(n, nb, d_out, x.device)
            gemm(x, basis, out=xb_nb[:n].permute(1, 0, 2))
            graph.pair_stamp += 1                                 # (the pair buffers are rewritten here)

            def pair_route():
                stream_gather(pair, att, write_zeros=False, out=cells.view(-1, nb)[:n * n], label='pair_cells[dd.fwd]', kind=1)
                return sum_slabs(pair_product(cells, xb_nb, symmetric=pair.symmetric).view(-1, n, d_out))
            routes.append(('pair', pair_route))
        times = {'pair': float('inf'), 'y': 0.0}
        for name, fn in routes:
            fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(3):
                fn()
            b.record()
            torch.cuda.synchronize(x.device)
            times[name] = a.elapsed_time(b) / 3 * 1e3
        total = torch.tensor([min(times['pair'], 1e30), times['y']], dtype=torch.float32, device=x.device)
        shard.all_reduce(total)                                   # the same two sums on every rank -> the same decision
        total = total.tolist()
    choice = 'pair' if total[0] <= total[1] else 'y'
    graph.fwd_route[key] = (choice, times, {'pair': total[0], 'y': total[1]})
    return choice if pair is not None else 'y'


class _RGCN(torch.autograd.Function):
    """One basis-decomposed R-GCN layer with global-mean aggregation
    (reference MyRGCNConv2 / MyRGCNConv, src/layers.py:102-193 / :21-99):

        out = D^-1 sum_r A_r X W_r + X root,   W_r = sum_b att[r,b] basis[b]

    evaluated basis-first / transform-then-gather: XB_b = X basis_b, Y = att . XB, then one
    gather-sum over the rows of Y (no E x d intermediates).

    With a `shard` (tip_amd/dist.py) the graph, `att` and its gradient hold only this rank's relations
    (`graph.scale` = 1 / GLOBAL in-degree): the partial aggregate is all-reduced before the 1/deg
    scaling, and [partial dX | partial d basis] are all-reduced as ONE flat buffer that the GEMMs write
    into directly; d att rows are shard-local and never travel."""

    @staticmethod
    def forward(ctx, x, basis, att, root, graph, shard, relu, gate_input=False, defer_output=False, partner=None, cells_token=None):
        """relu: False | True | 'gated_downstream' (ReLU applied here, its backward mask applied by the
        consumer, which must be the ONLY consumer and run with gate_input=True).
        gate_input: x is the ReLU output of the producing layer; dX is masked with (x > 0) while it is
        finished, i.e. the producer's ReLU backward is fused into this layer's last kernel."""
        x, basis, att, root = _f32c(x), basis.contiguous(), att.contiguous(), root.contiguous()
        n, d_in = x.shape
        nb, _, d_out = basis.shape
        r = att.shape[0]
        # x may be the output of the previous layer with its final slab sum still PENDING (`defer_output`): this layer then
        # finishes it in the launch that computes its own XB / X root (`sum_slabs_xb`), or right here if it takes another route
        pend = x.__dict__.pop('_tipk_pending', None) if hasattr(x, '__dict__') else None
        pair = graph.pair_fwd if r > 0 else None
        if pair is not None and pair.symmetric and not lib().tipk_pair_product_supported(nb, d_out):
            pair = None                                      # only the dedicated product kernel reads mirrored cells
        if pair is not None and not (pair.n_table == r and pair.n_rows == n * n and stream_gather_split(r, nb)
                                     and (nb // stream_gather_split(r, nb)) // 4 == pair.lanes):
            pair = None
        if shard is not None and shard.world > 1:
            # the pair form's dense product reads the whole N x N cell matrix on EVERY rank (16 us at BioSNAP, whatever
            # the rank's share of the relations); the Y route (Y = att . XB + unit gather) scales with the share.  One
            # decision for the whole job, the same from run to run unless timing is asked for; every rank asks, whether or
            # not it could take the pair form itself (`_fwd_route`: the timed mode's all-reduce is a collective)
            if _fwd_route(graph, x, basis, att, pair, shard) == 'y':
                pair = None
        if pend is not None and not (pair is not None and shard is None and any(ctx.needs_input_grad[:3]) and pair_grads_supported(nb, d_out)
                                     and graph.pair_bwd is not None and sum_slabs_xb_supported(d_in, d_out)):
            _finish_pending(x, pend)
            pend = None
        if pair is not None:
            # PAIR FORM.  sum_r A_r X W_r = sum_{(u -> v)} sum_b C[v, u, b] XB_b[u],  C[v, u, :] = sum of att[r, :] over
            # the relations r that link u -> v.  A BioSNAP drug pair is linked by 66 relations on average, so the
            # per-EDGE work shrinks to adding one att row (LDS-resident for the whole launch: a wave-stream gather
            # like the transposed pass) and the per-PAIR work is one dense product -- Y = att . XB [R N, out]
            # (91 MB written and gathered back at layer 1) is never formed.
            # cells are kept SOURCE-major, C[u, v, :]: the product  sum_u C[u] (N x bases) . XB[u] (bases x out)  then reads
            # 16 KB contiguous per operand tile (destination-major cells made every tile 128 separate 128-byte rows:
            # 2.5 TB/s); the u range is padded to a multiple of PAIR_KGROUP with cells / XB rows that stay zero
            cells, xb_nb, zeros = graph.pair_buffers(n, nb, d_out, x.device)
            need_grad = any(ctx.needs_input_grad[:3])
            ctx.pair_bwd = need_grad and pair_grads_supported(nb, d_out) and graph.pair_bwd is not None
            rs = None if (ctx.pair_bwd or not need_grad) else graph.rs_bwd
            # every pass through here rewrites the graph's cell / XB buffers: a backward pass that finds another stamp than
            # its own forward pass left knows they are no longer its operands
            graph.pair_stamp += 1
            if ctx.pair_bwd:
                # PAIR-FORM BACKWARD (round 5; include/tipk.h section 2e): the backward pass reads the cells and the node-major
                # XB this pass leaves in the graph's buffers -- nothing else is saved, nothing is transposed
                if pend is not None:                         # x itself + XB + X root in ONE launch (layer hand-over)
                    xroot = sum_slabs_xb(pend[0], pend[1], pend[2], pend[3], x, basis, root, xb_nb)
                    pend = None
                else:
                    _, xroot = gemm_group([gemm_job(x, basis, out=xb_nb[:n].permute(1, 0, 2)), gemm_job(x, root)])
                xb, xbt = None, None
                ctx.xb_stamp = graph.pair_stamp
            elif rs is not None and rs.compact is not None:
                # XB is computed ONCE, into the node-major buffer the pair product reads; the backward pass
                # (`node_products` on the compact dY) takes the same buffer through strides -- a node's 32 rows are one
                # 4-KB block there.  The buffer belongs to the graph and the next forward pass rewrites it: the stamp tells
                # a backward pass that runs after ANOTHER forward to recompute XB instead of reading someone else's.
                _, xroot = gemm_group([gemm_job(x, basis, out=xb_nb[:n].permute(1, 0, 2)), gemm_job(x, root)])
                # ... and the pair product, which stages every node's XB block in LDS anyway, writes it back a second time as
                # [N, out, bases]: what the d att product of the backward pass reads (`node_products` xbt)
                xbt = graph.xbt_buffer(n, nb, d_out, x.device)
                xb = xb_nb[:n].permute(1, 0, 2)
                ctx.xb_stamp = graph.pair_stamp
            else:
                xb, _, xroot = gemm_group([gemm_job(x, basis), gemm_job(x, basis, out=xb_nb[:n].permute(1, 0, 2)), gemm_job(x, root)])
                ctx.xb_stamp, xbt = None, None
            # the token is an OBJECT the caller made for this forward pass and handed to both layers (FMEncoder.forward): it says
            # "the previous layer gathered your cells in THIS pass" -- not a (pointer, version) pair, which a raw-pointer parameter
            # update (tipk_adam_step does not bump `_version`) would leave looking current
            tok = getattr(graph, 'cells_token', None)
            graph.cells_token = None
            if cells_token is not None and tok is not None and tok[0] is cells_token and tok[1] == graph.pair_stamp - 1:
                pass                                         # this layer's cells were gathered with the previous layer's (below)
            elif partner is not None and shard is None and pair_cells_partner_ok(graph, att, partner[1], partner[0], n):
                # the cells depend on the parameters only: the NEXT layer's (same graph, its own att) are gathered in this
                # launch too -- one launch ramp and one tail instead of two.  The partner's buffers are rewritten: its stamp
                # moves on, and it is told for which att (object, version) and stamp its cells are current
                att2, g2, d_out2 = partner[0].contiguous(), partner[1], partner[2]
                cells2 = g2.pair_buffers(n, nb, d_out2, x.device)[0]
                stream_gather_two(pair, att, att2, cells.view(-1, nb)[:n * n], cells2.view(-1, nb)[:n * n])
                g2.pair_stamp += 1
                g2.cells_token = (partner[3] if len(partner) > 3 else None, g2.pair_stamp)
            else:
                stream_gather(pair, att, write_zeros=False, out=cells.view(-1, nb)[:n * n], label='pair_cells[dd.fwd]', kind=1)
            slabs = pair_product(cells, xb_nb, symmetric=pair.symmetric, links=getattr(pair, 'links', None), zeros=zeros, xbt=xbt)
            if shard is None and defer_output and relu == 'gated_downstream' and d_out == 32:
                # the ordered slab sum is left to the ONE consumer of this output (the next R-GCN layer, which runs it in the
                # launch of its own XB product): until then `out` is storage only
                out = torch.empty((n, d_out), dtype=torch.float32, device=x.device)
                out._tipk_pending = (slabs.view(-1, n, d_out), graph.scale, xroot, True)
            elif shard is None:
                out = sum_slabs(slabs.view(-1, n, d_out), row_scale=graph.scale, addend=xroot, relu=bool(relu))
            else:
                agg = sum_slabs(slabs.view(-1, n, d_out))
                shard.all_reduce(agg)
                out = sum_slabs(agg.view(1, n, d_out), row_scale=graph.scale, addend=xroot, relu=bool(relu))
            ctx.graph, ctx.shard, ctx.relu, ctx.gate_input = graph, shard, relu, gate_input
            ctx.save_for_backward(x, basis, att, root, xb if ctx.xb_stamp is None else None, out if relu is True else None)
            return out
        ctx.xb_stamp, ctx.pair_bwd = None, False
        use_rl = r > 0 and rel_gather_usable(graph.rl_fwd, n, d_out, False)   # (the unit plan is built here, on first use)
        # LARGE node sets (round 5; include/tipk.h sections 2h / 2f): Y = att . XB [R N, out] is never formed -- the (relation,
        # node) row sums are multiplied where they are assembled, or per destination T[:, v, :] = sum_e att[r_e, :]^T (x) X[src_e]
        # (a product over the node's incoming edges); then sum_b T_b basis_b
        rows = _row_plan(graph, False, n, r, nb, d_in) if (r > 0 and not use_rl) else None
        dest = graph.dest_fwd if (r > 0 and not use_rl and rows is None) else None
        if use_rl:
            assert graph.rl_fwd.n_nodes == n and graph.rl_fwd.n_rel == r, 'graph/plan mismatch'
        elif rows is not None:
            assert rows.n_nodes == n and rows.n_rel == r, 'graph/plan mismatch'
        elif r > 0 and dest is None:                                         # (the generic plan is built here, on first use)
            assert graph.fwd.n_out == n and graph.fwd.n_table == r * n, 'graph/plan mismatch'
        xb, xroot = gemm_group([gemm_job(x, basis), gemm_job(x, root)])      # XB and X root: one grouped launch
        if dest is not None or rows is not None:
            # (row sums first: 0.16 TFLOP per layer at config 5 instead of the 0.41 of the per-edge product)
            t_b = row_products(rows, x, att) if rows is not None else dest_products(dest, x, att)
            agg = gemm(t_b, basis, reduce_batch=True, kgroup=large_kgroup(n, nb))
            if shard is not None:
                shard.all_reduce(agg)
            out = sum_slabs(agg.view(1, n, d_out), row_scale=graph.scale, addend=xroot, relu=bool(relu))
            ctx.graph, ctx.shard, ctx.relu, ctx.gate_input = graph, shard, relu, gate_input
            ctx.save_for_backward(x, basis, att, root, xb, out if relu is True else None)
            return out
        y = gemm(att, xb.view(nb, n * d_out)).view(r * n, d_out) if r > 0 else None     # [R N, out]
        if shard is None:
            if use_rl:
                # LDS-resident gather -> per-workgroup partial slabs; the ordered slab sum also applies
                # 1/deg, adds X root and the ReLU: the layer is finished in one pass
                part = rel_gather(graph.rl_fwd, y, backward=False, reduce=False)
                out = sum_slabs(part, row_scale=graph.scale, addend=xroot, relu=bool(relu))
            else:
                agg = gather_sum(graph.fwd, y, row_scale=graph.scale)
                out = sum_slabs(agg.view(1, n, d_out), addend=xroot, relu=bool(relu))
        else:
            # partial aggregate of this rank's relations -> one all-reduce -> 1/deg(global), + X root, ReLU
            if r == 0:
                agg = torch.zeros((n, d_out), dtype=torch.float32, device=x.device)
            elif use_rl:
                agg = sum_slabs(rel_gather(graph.rl_fwd, y, backward=False, reduce=False))
            else:
                agg = gather_sum(graph.fwd, y)
            shard.all_reduce(agg)
            out = sum_slabs(agg.view(1, n, d_out), row_scale=graph.scale, addend=xroot, relu=bool(relu))
        del y
        ctx.graph, ctx.shard, ctx.relu, ctx.gate_input = graph, shard, relu, gate_input
        ctx.save_for_backward(x, basis, att, root, xb, out if relu is True else None)
        return out

    @staticmethod
    def backward(ctx, g):
        x, basis, att, root, xb, out = ctx.saved_tensors
        graph, shard = ctx.graph, ctx.shard
        g = _f32c(g).contiguous()
        if ctx.relu is True:
            g = rows_affine(g, gate=out)                                 # ReLU gate of the fused epilogue
        n, d_in = x.shape
        nb, _, d_out = basis.shape
        r = att.shape[0]
        xbt = None
        j_att = None
        if ctx.pair_bwd and r > 0:
            cells, xb_nb, _ = graph.pair_buffers(n, nb, d_out, x.device)
            if graph.pair_stamp != ctx.xb_stamp:
                # another forward pass has rewritten the graph's buffers: the same values again -- and a new stamp, so that
                # THAT pass's backward pass recomputes its own as well
                pair = graph.pair_fwd
                gemm(x, basis, out=xb_nb[:n].permute(1, 0, 2))
                stream_gather(pair, att, write_zeros=False, out=cells.view(-1, nb)[:n * n], label='pair_cells[dd.fwd]', kind=1)
                graph.pair_stamp += 1
                ctx.xb_stamp = graph.pair_stamp
            j_att, g_xb = pair_backward(graph.pair_bwd, cells, xb_nb, g)
        elif ctx.xb_stamp is not None:                                   # XB lives in the graph's node-major buffer (forward)
            if graph.pair_stamp == ctx.xb_stamp:
                xb = graph.pair_buffers(n, nb, d_out, x.device)[1][:n].permute(1, 0, 2)
                xbt = graph.xbt_buffer(n, nb, d_out, x.device)
            else:                                                        # another forward pass has rewritten it: same values again
                xb = gemm(x, basis)
        xb2 = xb.view(nb, n * d_out) if (xb is not None and xb.is_contiguous()) else None
        if j_att is not None:
            pass
        elif r > 0:
            rs = graph.rs_bwd
            used = None
            if rs is not None and rs.compact is not None:
                assert rel_stream_split(n, d_out) and (d_out // rel_stream_split(n, d_out)) // 4 == rs.lanes and rs.n_rel == r
                # dY_r = A_r^T (D^-1 g) in COMPACT node-major form: only the (relation, source) rows that have an edge
                # exist; both products of dY run on that form (d XB complete, d att as a few small slabs)
                dyc = rel_stream_bwd(rs, g, row_scale=graph.scale)
                j_att, g_xb = node_products(dyc, rs.compact, att, xb, xbt)
            elif rs is not None and rel_stream_split(n, d_out) and (d_out // rel_stream_split(n, d_out)) // 4 == rs.lanes:
                # dY_r = A_r^T (D^-1 g), 1/deg fused.  Rows (relation, node) without edges -- half of them -- are
                # neither written here nor read as data by the fused products (row mask)
                masked = dy_products_fused(r, n * d_out, nb) and xb.stride(-1) == 1
                g_y = rel_stream_bwd(rs, g, row_scale=graph.scale, write_zeros=not masked).view(r, n * d_out)
                used = rs.row_used if masked else None
            elif rel_gather_usable(graph.rl_bwd, n, d_out, True):
                g_y = rel_gather(graph.rl_bwd, g, backward=True, row_scale=graph.scale).view(r, n * d_out)
            elif xb2 is not None and n * d_out * 4 < 2 ** 31 and _row_plan(graph, True, n, r, nb, d_out) is not None:
                # LARGE node sets: dY = A_r^T (D^-1 g') is never written -- its rows are summed in LDS and multiplied there
                gs = rows_affine(g, row_mul=graph.scale)
                j_rows, g_xb = row_products(graph.row_bwd, gs, att, xb2)
                gemm_group([], [j_rows])                                 # (thousands of slabs: a slab sum of its own, as dy_products')
                g_att, g_y = j_rows.out, None
            else:
                gs = rows_affine(g, row_mul=graph.scale)
                csr = graph.csr_bwd if (d_out % 4 == 0 and 8 <= d_out <= 256 and gs.numel() * 4 < 2 ** 32) else None   # (32-bit row offsets)
                if csr is not None:                                      # R N short rows: contiguous streams, no descriptors
                    g_y = gather_rows_csr(csr, gs).view(r, n * d_out)
                else:
                    g_y = gather_sum(graph.bwd, gs).view(r, n * d_out)
            if j_att is None and g_y is not None:
                # both consumers of dY in one pass over it (+ one grouped slab sum)
                g_att, g_xb = dy_products(g_y, att, xb2, used, n)
                g_xb = g_xb.view(nb, n, d_out)
        else:
            g_att = torch.zeros((0, nb), dtype=torch.float32, device=x.device)
            g_xb = torch.zeros((nb, n, d_out), dtype=torch.float32, device=x.device)
        extra = [] if j_att is None else [j_att]                         # the d att slabs ride in this layer's grouped slab sum
        if j_att is not None:
            g_att = j_att.out
        # d basis, d root and both halves of dX are independent given dXB and g: one grouped launch.  When the reductions
        # fit the waves of one workgroup per output tile (BioSNAP: 645 and 1 056 terms) the products are FINISHED by that
        # launch, together with the d att slab sum (round 4: 10.4 + 7.6 us of split-K slabs + their sum per layer before)
        if len(extra) <= _lib.WG_SUMS_MAX:
            flat = None
            if shard is not None:
                flat = torch.empty(n * d_in + nb * d_in * d_out, dtype=torch.float32, device=x.device)
            w_basis = wg_gemm_job(x.t(), g_xb, out=None if flat is None else flat[n * d_in:].view(nb, d_in, d_out))
            w_root = wg_gemm_job(x.t(), g)
            if shard is None:
                w_x = wg_gemm_job(g_xb, basis.transpose(1, 2), reduce_batch=True, a2=g, b2=root.t(),
                                  gate=x if ctx.gate_input else None)
            else:
                w_x = wg_gemm_job(g_xb, basis.transpose(1, 2), reduce_batch=True, out=flat[:n * d_in].view(n, d_in))
            if w_basis is not None and w_root is not None and w_x is not None:
                wg_gemm_group([w_basis, w_root, w_x], extra)
                g_x = w_x.out
                if shard is not None:
                    shard.all_reduce(flat)
                    g_x = gemm(g, root.t(), out=g_x, c_in=g_x)           # replicated term, added once
                    if ctx.gate_input:
                        g_x = rows_affine(g_x, gate=x)
                return g_x, w_basis.out, g_att, w_root.out, None, None, None, None, None, None, None
        j_root = gemm_job(x.t(), g)
        if shard is None:
            j_basis = gemm_job(x.t(), g_xb)                              # [B, in, out]
            j_xr = gemm_job(g, root.t(), ksplit=1)                       # written by the GEMM launch itself: the
            g_x = j_xr.out                                               # basis half is summed on top of it afterwards
            j_xq = gemm_job(g_xb, basis.transpose(1, 2), out=g_x, c_in=g_x, reduce_batch=True, kgroup=large_kgroup(n, nb))
            if j_xq.slabs is not None:                                   # summed on top of g root^T afterwards
                if ctx.gate_input:
                    j_xq.gate = x                                        # ... and masked with (x > 0) in the same pass
                gemm_group([j_basis, j_root, j_xr, j_xq], extra)
            else:                                                        # reads g_x while accumulating
                gemm_group([j_basis, j_root, j_xr], extra)
                gemm_group([j_xq])
                if ctx.gate_input:
                    g_x = rows_affine(g_x, gate=x)
            g_basis, g_root = j_basis.out, j_root.out
        else:
            # [partial dX | partial d basis] live in ONE flat buffer the products write into: a single
            # all-reduce, no pack / unpack copies; d att (this rank's rows) and d root (replicated
            # operands: identical on every rank) need no collective
            flat = torch.empty(n * d_in + nb * d_in * d_out, dtype=torch.float32, device=x.device)
            g_x = flat[:n * d_in].view(n, d_in)
            g_basis = flat[n * d_in:].view(nb, d_in, d_out)
            j_basis = gemm_job(x.t(), g_xb, out=g_basis)
            j_xq = gemm_job(g_xb, basis.transpose(1, 2), out=g_x, reduce_batch=True, kgroup=large_kgroup(n, nb))   # partial over this shard
            gemm_group([j_basis, j_root, j_xq], extra)
            g_root = j_root.out
            shard.all_reduce(flat)
            g_x = gemm(g, root.t(), out=g_x, c_in=g_x)                   # replicated term, added once
            if ctx.gate_input:
                g_x = rows_affine(g_x, gate=x)
        return g_x, g_basis, g_att, g_root, None, None, None, None, None, None, None


def rgcn(x, basis, att, root, graph, shard=None, relu=False, gate_input=False, defer_output=False, partner=None, cells_token=None):
    """partner (optional): (att, graph, d_out, token) of the NEXT R-GCN layer on the same D-D graph -- its pair cells are gathered
    in this layer's cell launch (`stream_gather_two`); the next layer is then called with cells_token = the same token object."""
    return _RGCN.apply(x, basis, att, root, graph, shard, relu, gate_input, defer_output, partner, cells_token)


class _DrugMix(torch.autograd.Function):
    """x0 = cat(embed / d_norm, pd) or embed / d_norm + pd  (src/layers.py:532-539)."""

    @staticmethod
    def forward(ctx, xd, pd, d_norm, cat):
        xd, pd = _f32c(xd), _f32c(pd)
        n, ne = xd.shape
        if cat:
            out = torch.empty((n, ne + pd.shape[1]), dtype=torch.float32, device=xd.device)
            rows_affine(xd, row_div=d_norm, out=out[:, :ne])
            rows_affine(pd, out=out[:, ne:])
        else:
            out = rows_affine(xd, row_div=d_norm)
            rows_affine(pd, out=out, accumulate=True)
        ctx.cat, ctx.ne = cat, ne
        ctx.save_for_backward(d_norm)
        return out

    @staticmethod
    def backward(ctx, g):
        (d_norm,) = ctx.saved_tensors
        g = _f32c(g)
        ne = ctx.ne
        if ctx.cat:
            return rows_affine(g[:, :ne], row_div=d_norm), rows_affine(g[:, ne:]), None, None
        return rows_affine(g, row_div=d_norm), g, None, None


def drug_mix(xd, pd, d_norm, cat):
    return _DrugMix.apply(xd, pd, d_norm, cat)


class _DrugMixMM(torch.autograd.Function):
    """x0 = cat(embed / d_norm, mean @ W) or embed / d_norm + mean @ W: the dense map of
    MyHierarchyConv (src/layers.py:239) writes straight into the mixed feature matrix
    (:532-539), and its gradient is read from a column-slice view on the way back."""

    @staticmethod
    def forward(ctx, xd, mean, weight, d_norm, cat):
        xd, mean, weight = _f32c(xd), _f32c(mean), _f32c(weight)
        n, ne = xd.shape
        pd_dim = weight.shape[1]
        p = weight.shape[0]
        ctx.fused = p <= 64 and pd_dim <= 64 and weight.is_contiguous() and d_norm.is_contiguous()
        if ctx.fused:                                                     # scaling, cat | add and the dense map: one launch
            out = torch.empty((n, ne + pd_dim if cat else ne), dtype=torch.float32, device=xd.device)
            check(lib().tipk_drug_mix_fwd(ptr(xd), xd.stride(0), ptr(d_norm), ptr(mean), mean.stride(0), ptr(weight), p, pd_dim,
                                          n, ne, int(cat), ptr(out), out.stride(0), stream_ptr(xd.device)), 'tipk_drug_mix_fwd')
            ctx.cat, ctx.ne = cat, ne
            ctx.save_for_backward(mean, weight, d_norm)
            return out
        if cat:
            out = torch.empty((n, ne + pd_dim), dtype=torch.float32, device=xd.device)
            rows_affine(xd, row_div=d_norm, out=out[:, :ne])
            gemm(mean, weight, out=out[:, ne:])
        else:
            out = rows_affine(xd, row_div=d_norm)
            gemm(mean, weight, out=out, c_in=out)
        ctx.cat, ctx.ne = cat, ne
        ctx.save_for_backward(mean, weight, d_norm)
        return out

    @staticmethod
    def backward(ctx, g):
        mean, weight, d_norm = ctx.saved_tensors
        g = _f32c(g)
        g_pd = g[:, ctx.ne:] if ctx.cat else g
        g_xd = rows_affine(g[:, :ctx.ne] if ctx.cat else g, row_div=d_norm)
        j_w = gemm_job(mean.t(), g_pd)
        j_m = gemm_job(g_pd, weight.t()) if ctx.needs_input_grad[1] else None
        gemm_group([j for j in (j_w, j_m) if j is not None])
        return g_xd, (j_m.out if j_m else None), j_w.out, None, None


def drug_mix_mm(xd, mean, weight, d_norm, cat):
    return _DrugMixMM.apply(xd, mean, weight, d_norm, cat)


def drug_mix_gather_supported(h, weight, d_norm):
    p, q = weight.shape
    return bool(h.is_cuda and h.dtype == torch.float32 and h.stride(1) == 1 and weight.is_contiguous() and
                (d_norm is None or d_norm.is_contiguous()) and lib().tipk_drug_mix_gather_supported(int(p), int(q)))


def drug_mix_gather_launch(xd, h, weight, d_norm, cat, graph):
    """(x0, mean, weight as passed to the kernel): the forward launch of the fused P -> D + drug-mix stage (`tipk_drug_mix_gather_fwd`)."""
    xd, h, weight = _f32c(xd), _f32c(h), _f32c(weight).contiguous()
    csr = graph.pd_csr
    require_device(xd, h, weight, d_norm, csr['fwd_ptr'])
    n, ne = xd.shape
    p, q = weight.shape
    assert h.shape == (csr['n_src'], p) and csr['fwd_ptr'].numel() == n + 1
    out = torch.empty((n, ne + q if cat else ne), dtype=torch.float32, device=xd.device)
    mean = torch.empty((n, p), dtype=torch.float32, device=xd.device)
    with _timed('drug_mix_gather_fwd[%dx%dx%d]' % (n, p, q)):
        check(lib().tipk_drug_mix_gather_fwd(ptr(xd), xd.stride(0), ptr(d_norm), ptr(h), h.stride(0), ptr(csr['fwd_ptr']),
                                             ptr(csr['fwd_src']), ptr(csr['scale']), ptr(csr['fwd_wg']), ptr(csr['fwd_order']), csr['fwd_wg'].shape[0],
                                             ptr(weight), p, q, n, ne, int(cat),
                                             ptr(out), out.stride(0), ptr(mean), stream_ptr(xd.device)),
              'tipk_drug_mix_gather_fwd')
    return out, mean, weight


class _DrugMixGather(torch.autograd.Function):
    """x0 = cat(xd / d_norm, mean(H) W) or xd / d_norm + mean(H) W with mean(H)[d] = the mean of the source rows that point
    at d (MyHierarchyConv, src/layers.py:229-242, + the mix of :532-539): one launch forward; backward one launch for
    d xd, d mean and d W + the transposed gather of d mean on its plan (include/tipk.h section 3, csrc/tipk_drugmix.hip)."""

    @staticmethod
    def forward(ctx, xd, h, weight, d_norm, cat, graph):
        out, mean, weight = drug_mix_gather_launch(xd, h, weight, d_norm, cat, graph)
        ctx.cat, ctx.ne, ctx.graph = cat, xd.shape[1], graph
        ctx.save_for_backward(mean, weight, d_norm)
        return out

    @staticmethod
    def backward(ctx, g):
        mean, weight, d_norm = ctx.saved_tensors
        g = _f32c(g)
        n, p = mean.shape
        q = weight.shape[1]
        dev = g.device
        g_xd = torch.empty((n, ctx.ne), dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        g_mean = torch.empty((n, p), dtype=torch.float32, device=dev) if ctx.needs_input_grad[1] else None
        g_w = torch.empty((p, q), dtype=torch.float32, device=dev)
        with _timed('drug_mix_bwd[%dx%dx%d]' % (n, p, q)):
            check(lib().tipk_drug_mix_bwd(ptr(g), g.stride(0), ptr(d_norm), ptr(mean), ptr(weight), p, q, n, ctx.ne, int(ctx.cat),
                                          ptr(g_xd), g_xd.stride(0) if g_xd is not None else 0, ptr(g_mean), ptr(g_w),
                                          stream_ptr(dev)), 'tipk_drug_mix_bwd')
        # back to the source rows: the transposed plan carries 1 / count of the edge's target as edge weights
        g_h = gather_sum(ctx.graph.bwd, g_mean) if g_mean is not None else None
        return g_xd, g_h, g_w, None, None, None


def drug_mix_gather(xd, h, weight, d_norm, cat, graph):
    return _DrugMixGather.apply(xd, h, weight, d_norm, cat, graph)


_ONES = {}


def _ones(n, device):
    key = (int(n), str(device))
    if key not in _ONES:
        _ONES[key] = torch.ones((1, int(n)), dtype=torch.float32, device=device)
    return _ONES[key]


class _GCNConv(torch.autograd.Function):
    """out = relu?(A_hat (x W^T) + bias): GCNConv as one node of the autograd graph, so that the
    three gradients that only need the transposed aggregate -- d W, d x and d bias (= 1^T g as a
    1-row product) -- leave in ONE grouped launch (PyG GCNConv.forward; src/layers.py:392-394)."""

    @staticmethod
    def forward(ctx, x, weight, bias, graph, relu, link=None):
        """relu: False | True | 'gated_downstream' -- the ReLU is applied here, its backward mask AND the column sums of the
        masked gradient (d bias, as partial rows in `link.parts`) are produced by the ONE consumer of the output, which runs
        with gate_input=True and the same `link` (`_GCNConvAggFirst`: in the epilogue of its transposed aggregation)."""
        ctx.identity = x is None
        if x is None:
            xl = transpose(weight)                                       # lin(I) = W^T
        else:
            x = _f32c(x)
            xl = gemm(x, weight.t())
        out = gather_sum(graph.fwd, xl, row_scale=graph.scale, bias=bias, relu=bool(relu))
        ctx.graph, ctx.relu, ctx.has_bias, ctx.link = graph, relu, bias is not None, link
        ctx.save_for_backward(x, weight, out if relu is True else None)
        return out

    @staticmethod
    def backward(ctx, g):
        x, weight, out = ctx.saved_tensors
        graph = ctx.graph
        g = _f32c(g).contiguous()
        bias_parts = None
        handed = ctx.relu == 'gated_downstream'                         # g arrives masked; its column sums may come with it
        if handed:
            g_pre = g
            bias_parts = ctx.link.parts if (ctx.link is not None and ctx.has_bias) else None
            if ctx.link is not None:
                ctx.link.parts = None
        elif ctx.relu and ctx.has_bias:                                 # ReLU gate + stage 1 of d bias in one pass
            fused = gate_colsum(g, out)
            if fused is not None:
                g_pre, bias_parts = fused
        if bias_parts is None and not handed:
            g_pre = rows_affine(g, gate=out) if ctx.relu else g
        g_agg = rows_affine(g_pre, row_mul=graph.scale) if graph.scale is not None else g_pre
        s_bias = None
        if ctx.identity and bias_parts is not None:
            # identity features: nothing else is summed after the transposed aggregation -- the bias partials ride in ITS launch
            s_bias = slab_job(bias_parts)
            g_table = gather_sum(graph.bwd, g_agg, riders=[s_bias])
        else:
            g_table = gather_sum(graph.bwd, g_agg)
        j_b = None
        if ctx.has_bias and bias_parts is None:
            j_b = gemm_job(_ones(g_pre.shape[0], g.device), g_pre)
        if ctx.identity:
            # d W = (d lin)^T: a VIEW when the parameter is stored transposed (tip_amd.layers._Lin), its strides
            # then equal the parameter's, so the optimizer's fused / foreach paths apply
            g_w = g_table.t() if weight.t().is_contiguous() else transpose(g_table)
            if s_bias is not None:
                return None, g_w, s_bias.out.view(-1), None, None, None
            if j_b is not None:
                gemm_group([j_b])
            return None, g_w, (j_b.out.view(-1) if j_b else None), None, None, None
        # d W in the parameter's own memory layout (tip_amd.layers._Lin keeps [in, out] storage behind the [out, in]
        # shape): autograd's AccumulateGrad otherwise re-lays the gradient out with a copy kernel of its own
        w_t = weight.t().is_contiguous() and not weight.is_contiguous()
        j_w = gemm_job(x.t(), g_table) if w_t else gemm_job(g_table.t(), x)
        g_w = j_w.out.t() if w_t else j_w.out
        j_x = gemm_job(g_table, weight) if ctx.needs_input_grad[0] else None
        if bias_parts is not None:                                      # rides in the grouped slab sum below
            s_b = slab_job(bias_parts)
            gemm_group([j for j in (j_w, j_x) if j is not None], [s_b])
            return (j_x.out if j_x else None), g_w, s_b.out.view(-1), None, None, None
        gemm_group([j for j in (j_w, j_x, j_b) if j is not None])
        return (j_x.out if j_x else None), g_w, (j_b.out.view(-1) if j_b else None), None, None, None


class GateLink(object):
    """Hand-over between a ReLU layer (`relu='gated_downstream'`) and the one consumer of its output (`gate_input=True`): the
    consumer's backward pass leaves the column sums of the masked gradient here (partial rows [W, 1, d]) for the layer's."""
    __slots__ = ('parts',)

    def __init__(self):
        self.parts = None


def gcn_conv(x, weight, bias, graph, relu=False, link=None):
    """x = None means identity features."""
    return _GCNConv.apply(x, weight, bias, graph, relu, link)


class _GCNConvAggFirst(torch.autograd.Function):
    """out = relu?((A_hat x) W^T + bias) -- the same map as `_GCNConv` (A_hat (x W^T) = (A_hat x) W^T), aggregate FIRST: for a
    layer whose output is needed for a few rows only (`gcn_norm_graph(rows=...)`: conv2 of the P-P encoder, 3 640 of 19 081
    proteins) the dense map then runs on the kept rows inside the gather's launch (`gather_sum_lin`), and the backward pass
    is one launch for g W, d W = g^T (A_hat x) and d bias (reductions over the kept rows: `wg_gemm_group`) + the transposed
    gather -- 1 + 2 launches where transform-first took 2 + 3 (src/layers.py:392-394)."""

    @staticmethod
    def forward(ctx, x, weight, bias, graph, relu, gate_input=False, link=None):
        """gate_input: x is the ReLU output of a layer called with relu='gated_downstream' -- dx is masked with (x > 0), and
        the column sums of the masked dx go to `link.parts`, in the epilogue of the transposed aggregation."""
        x = _f32c(x)
        agg, out = gather_sum_lin(graph.fwd, x, weight, bias, relu, row_scale=graph.scale)
        ctx.graph, ctx.relu, ctx.has_bias, ctx.gate_input, ctx.link = graph, relu, bias is not None, gate_input, link
        ctx.save_for_backward(agg, weight, out if relu else None, x if gate_input else None)
        return out

    @staticmethod
    def backward(ctx, g):
        agg, weight, out, x_in = ctx.saved_tensors
        graph = ctx.graph
        g = _f32c(g).contiguous()
        if ctx.relu:
            g = rows_affine(g, gate=out)
        # d W in the parameter's own memory layout (tip_amd.layers._Lin keeps [in, out] storage behind the [out, in] shape)
        w_t = weight.t().is_contiguous() and not weight.is_contiguous()
        j_w = wg_gemm_job(agg.t(), g) if w_t else wg_gemm_job(g.t(), agg)
        j_x = wg_gemm_job(g, weight) if ctx.needs_input_grad[0] else None
        j_b = wg_gemm_job(_ones(g.shape[0], g.device), g) if ctx.has_bias else None
        if j_w is None or (ctx.needs_input_grad[0] and j_x is None) or (ctx.has_bias and j_b is None):
            # reductions beyond one workgroup's reach: the grouped split-K products and their slab sum
            j_w = gemm_job(agg.t(), g) if w_t else gemm_job(g.t(), agg)
            j_x = gemm_job(g, weight) if ctx.needs_input_grad[0] else None
            j_b = gemm_job(_ones(g.shape[0], g.device), g) if ctx.has_bias else None
            # ... whose slab sums ride in the transposed aggregation's launch (it only needs g W, which is written directly)
            riders = gemm_group([j for j in (j_w, j_x, j_b) if j is not None], defer_sums=j_x is not None and j_x.slabs is None)
            if j_x is None or j_x.slabs is not None:
                riders = []
        else:
            wg_gemm_group([j for j in (j_w, j_x, j_b) if j is not None])
            riders = []
        g_w = j_w.out.t() if w_t else j_w.out
        g_x = None
        if j_x is not None:
            gw = j_x.out if graph.scale is None else rows_affine(j_x.out, row_mul=graph.scale)
            if ctx.gate_input and gather_sum_epilogue_supported(graph.bwd, gw.shape[1]):
                # the producer's ReLU backward and the partial rows of ITS bias gradient in this launch's epilogue
                want = ctx.link is not None
                res = gather_sum(graph.bwd, gw, riders=riders, gate=x_in, colsum=want)
                g_x = res[0] if want else res
                if want:
                    ctx.link.parts = res[1]
            else:
                g_x = gather_sum(graph.bwd, gw, riders=riders)
                if ctx.gate_input:
                    g_x = rows_affine(g_x, gate=x_in)
        return g_x, g_w, (j_b.out.view(-1) if j_b is not None else None), None, None, None, None


def gcn_conv_agg_first(x, weight, bias, graph, relu=False, gate_input=False, link=None):
    return _GCNConvAggFirst.apply(x, weight, bias, graph, relu, gate_input, link)


class _DistMult(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, weight, edge_index, edge_type, sigmoid):
        score = distmult_fwd(z, weight, edge_index, edge_type, sigmoid)
        ctx.sigmoid = sigmoid
        ctx.save_for_backward(z, weight, edge_index, edge_type, score)
        return score

    @staticmethod
    def backward(ctx, g):
        z, weight, edge_index, edge_type, score = ctx.saved_tensors
        g_z, g_w = distmult_bwd(g, score, z, weight, edge_index, edge_type, ctx.sigmoid)
        return g_z, g_w, None, None, None


def distmult(z, weight, edge_index, edge_type, sigmoid=True):
    return _DistMult.apply(z, weight, edge_index, edge_type, sigmoid)


_UNIT_GRADS = {}          # data_ptr -> tensor: upstream gradients known to be exactly 1 (`unit_grad`)


def unit_grad(device):
    """A 0-dim ones tensor for `loss.backward(gradient=...)` that the fused objective recognises by its address: the
    backward pass then hands out the gradients of the forward launch as they are (no seed-filling launch, no scaling
    launch).  tip_amd.train.GraphedTrainStep uses it; the tensor must never be written."""
    device = torch.device(device)
    for t in _UNIT_GRADS.values():
        if t.device == device:
            return t
    t = torch.ones((), dtype=torch.float32, device=device)
    _UNIT_GRADS[t.data_ptr()] = t
    return t


class _DistMultLoss(torch.autograd.Function):
    """loss of TIP.forward (src/layers.py:335-340) with gradients produced in the same pass."""

    @staticmethod
    def forward(ctx, z, weight, pos_index, neg_index, edge_type):
        need = z.requires_grad or weight.requires_grad
        loss, g_z, g_w = distmult_loss(z, weight, pos_index, neg_index, edge_type, need_grad=need)
        ctx.save_for_backward(g_z, g_w)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        g_z, g_w = ctx.saved_tensors
        if g_z is None:
            return None, None, None, None, None
        if g.data_ptr() in _UNIT_GRADS:                    # the caller vouches for an upstream gradient of exactly 1
            return g_z, g_w, None, None, None
        g_z, g_w = torch._foreach_mul([g_z, g_w], g)       # one launch for both
        return g_z, g_w, None, None, None


def distmult_objective(z, weight, pos_index, neg_index, edge_type):
    return _DistMultLoss.apply(z, weight, pos_index, neg_index, edge_type)


def rank_metrics(pos_score, neg_score, range_ptr, max_pairs):
    """fp64 [3, R] (AUPRC, AUROC, AP) per relation on device (include/tipk.h section 6), or None if a
    relation is too large for the single-workgroup sort."""
    pos_score, neg_score = _f32c(pos_score).contiguous(), _f32c(neg_score).contiguous()
    require_device(pos_score, neg_score, range_ptr)
    n_rel = range_ptr.numel() - 1
    out = torch.empty((3, n_rel), dtype=torch.float64, device=pos_score.device)
    st = lib().tipk_rank_metrics(ptr(pos_score), ptr(neg_score), ptr(range_ptr), n_rel, int(max_pairs), ptr(out),
                                 stream_ptr(pos_score.device))
    if st == -2:                                   # TIPK_EUNSUPPORTED
        return None
    check(st, 'tipk_rank_metrics')
    return out


class _PairTable(torch.autograd.Function):
    """score[e] = sigma(s1[u_e, r_e] + s2[v_e, r_e])  (include/tipk.h section 4b)."""

    @staticmethod
    def forward(ctx, s1, s2, edge_index, edge_type, sigmoid):
        s1, s2 = _f32c(s1).contiguous(), _f32c(s2).contiguous()
        require_device(s1, s2, edge_index, edge_type)
        assert s1.shape == s2.shape
        validate_triples(edge_index, edge_type, s1.shape[0], s1.shape[1])
        u, v = _uv(edge_index)
        et = edge_type.contiguous()
        n = u.numel()
        score = torch.empty((n,), dtype=torch.float32, device=s1.device)
        check(lib().tipk_pair_table_fwd(ptr(s1), ptr(s2), s1.shape[1], ptr(u), ptr(v), _idx_bytes(u), ptr(et),
                                        _idx_bytes(et), n, int(sigmoid), ptr(score), stream_ptr(s1.device)),
              'tipk_pair_table_fwd')
        ctx.sigmoid, ctx.shape = sigmoid, s1.shape
        ctx.save_for_backward(edge_index, et, score)
        return score

    @staticmethod
    def backward(ctx, g):
        edge_index, et, score = ctx.saved_tensors
        g = _f32c(g).contiguous()
        u, v = _uv(edge_index)
        g1 = torch.zeros(ctx.shape, dtype=torch.float32, device=g.device)
        g2 = torch.zeros(ctx.shape, dtype=torch.float32, device=g.device)
        check(lib().tipk_pair_table_bwd(ptr(g), ptr(score), ctx.shape[1], ptr(u), ptr(v), _idx_bytes(u), ptr(et),
                                        _idx_bytes(et), u.numel(), int(ctx.sigmoid), ptr(g1), ptr(g2),
                                        stream_ptr(g.device)), 'tipk_pair_table_bwd')
        return g1, g2, None, None, None


class _PairTableLoss(torch.autograd.Function):
    """The fused TIP objective on the TRANSPOSED score tables of the NNDecoder (`tipk_pair_table_loss`): loss and both table
    gradients in one launch, reproducible bit for bit."""

    @staticmethod
    def forward(ctx, s1t, s2t, pos_index, neg_index, edge_type):
        s1t, s2t = _f32c(s1t).contiguous(), _f32c(s2t).contiguous()
        require_device(s1t, s2t, pos_index, neg_index, edge_type)
        r, n = s1t.shape
        assert s2t.shape == (r, n) and n <= 65535
        validate_triples(pos_index, edge_type, n, r)
        facts = _facts(edge_type)
        blocks = facts.get('rel_blocks')
        if blocks is None:
            rels, counts = torch.unique_consecutive(edge_type, return_counts=True)
            if torch.unique(rels).numel() != rels.numel():
                raise _lib.TipkError('the fused objective needs the triples grouped by relation (src/utils.py:35-65)')
            cnt = torch.zeros(r, dtype=torch.int64, device=edge_type.device)
            cnt[rels.to(torch.int64)] = counts
            rel_ptr = torch.cat([cnt.new_zeros(1), torch.cumsum(cnt, 0)]).contiguous()
            order = torch.sort(cnt, descending=True, stable=True).indices.to(torch.int32).contiguous()
            blocks = facts['rel_blocks'] = (rel_ptr, order)
        rel_ptr, order = blocks
        pp = packed_pairs(pos_index, n)
        if getattr(neg_index, '_tipk_packed_pairs', False):
            npk = neg_index
            if not getattr(neg_index, '_tipk_sampled', False):
                validate_triples(unpack_pairs(neg_index), None, n, r)
        else:
            if not getattr(neg_index, '_tipk_sampled', False):
                validate_triples(neg_index, None, n, r)
            w = neg_index[0].to(torch.int64) | (neg_index[1].to(torch.int64) << 16)
            npk = torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32).contiguous()
        assert npk.numel() == pp.numel()
        need = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        parts = torch.empty((r, 2), dtype=torch.float64, device=s1t.device)
        g1 = torch.empty_like(s1t) if need else None
        g2 = torch.empty_like(s2t) if need else None
        with _timed('pair_table_loss[%dx%d,positions=%d]' % (r, n, pp.numel())):
            check(lib().tipk_pair_table_loss(ptr(s1t), ptr(s2t), s1t.stride(0), n, r, ptr(pp), ptr(npk), ptr(rel_ptr), ptr(order),
                                             pp.numel(), 1e-13, ptr(parts), ptr(g1), ptr(g2), stream_ptr(s1t.device)),
                  'tipk_pair_table_loss')
        ctx.save_for_backward(g1, g2)
        return (parts.sum() * (-1.0 / pp.numel())).to(torch.float32).view(1)

    @staticmethod
    def backward(ctx, g):
        g1, g2 = ctx.saved_tensors
        return g1 * g, g2 * g, None, None, None


def pair_table_loss_supported(n_nodes):
    """True if `tipk_pair_table_loss` takes a node set of this size (two table rows of a relation in LDS: 24 B per node)."""
    return 0 < int(n_nodes) * 24 <= 150 * 1024 and int(n_nodes) <= 65535


def pair_table_objective(s1t, s2t, pos_index, neg_index, edge_type):
    """loss [1] of the fused objective on the transposed score tables s1t / s2t [R, N] (include/tipk.h section 4b)."""
    return _PairTableLoss.apply(s1t, s2t, pos_index, neg_index, edge_type)


def pair_table_score(s1, s2, edge_index, edge_type, sigmoid=True):
    return _PairTable.apply(s1, s2, edge_index, edge_type, sigmoid)
