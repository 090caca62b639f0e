"""Relation-sharded multi-GPU execution of the TIP training step (one process per GPU, RCCL over xGMI).

The reference is single-GPU (`README.md:58`).  A D-D R-GCN layer is a sum over relations of
independent partial aggregates of the same N x d output, and the DistMult objective is a sum over
triples that are grouped by relation, so the whole step shards by relation id (BASELINE.json
north_star):

  * relations are dealt to ranks by a greedy longest-processing-time rule on their edge counts
    (sizes span 450 ... 51 546 edges); a rank holds ONLY its relations' edges (train and test), its
    rows of `rgcn*.att` and of `decoder.weight` -- `shard_data_dict`, `shard_state_dict`;
  * `basis`, `root`, `embed` and the P-P / P->D parameters are replicated; the P-P and P->D stages
    (< 2 % of the work) are computed redundantly on every rank, so their gradients are identical
    everywhere and need no collective;
  * collectives per training step (all SUM all-reduce, fp32), each on ONE flat buffer that the
    producing kernels write into directly (no pack / unpack copies):
        forward   per R-GCN layer: the partial aggregate  sum_{r in shard} A_r Y_r   [N, d_out]
                  (division by the GLOBAL in-degree, + X root and the ReLU follow the reduce);
                  the loss: one scalar (each rank's objective is weighted by its share of the triples);
        backward  d z [N, n_hid2] of the decoder; per R-GCN layer [partial dX | partial d basis];
    d att / d decoder.weight rows are shard-local and never travel; `gather_state_dict` assembles a
    full checkpoint when one is wanted.

At BioSNAP size the collectives are pure latency (41 ... 430 KB) next to ~100 us kernels, so
sub-linear scaling is expected there; the synthetic 50 M-edge graph is the scaling case.
`torch.distributed` backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests of this logic and by
the 2-ranks-on-one-GPU tests.
"""
import torch
import torch.distributed as dist

import os

LOCAL_ROWS = ('encoder.rgcn1.att', 'encoder.rgcn2.att', 'decoder.weight')     # parameters held as shard-local rows
PEER_WAIT_MS = int(os.environ.get('TIPK_PEER_TIMEOUT_MS', '60000'))           # flag-wait budget of the one-shot exchange after its self-test


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
    """This rank's share of the relations, the process group of the partial sums, and what the local
    layers need to know about the WHOLE graph: the in-degree over all relations (the scatter-mean
    denominator, src/layers.py:123 aggr='mean') and the total number of training triples (the
    objective's mean)."""

    def __init__(self, rel_ids, rank, world, group=None, n_relations=None):
        self.rel_ids = torch.as_tensor(rel_ids, dtype=torch.int64)
        self.rank, self.world, self.group = rank, world, group
        self.n_relations = n_relations            # relations of the whole graph
        self.in_degree = None                     # int64 [N]: D-D in-degree over ALL relations
        self.n_train_total = None                 # directed training edges over all ranks
        self.n_train_local = None
        self.direct = None                        # DirectExchange (one-shot all-reduce over peer-mapped mailboxes)
        self.direct_sizes = None                  # None: every buffer the exchange takes; else {numel: True | False} (choose_collective)
        self.collective_report = None

    def rel_ids_on(self, device):
        if self.rel_ids.device != device:
            self.rel_ids = self.rel_ids.to(device)
        return self.rel_ids

    def all_reduce(self, flat):
        """In-place SUM all-reduce of one flat buffer: the one-shot exchange over peer-mapped mailboxes when it was
        enabled (`enable_direct_exchange`) and the buffer fits, the process group's all-reduce (RCCL / gloo) otherwise."""
        if self.direct is not None and self.direct.takes(flat) and self._direct_for(flat.numel()):
            self.direct.all_reduce(flat)
        elif self.world > 1 or dist.is_initialized():
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        return flat

    def _direct_for(self, numel):
        if self.direct_sizes is None:
            return True
        hit = self.direct_sizes.get(int(numel))
        if hit is not None:
            return hit
        won = [n for n, use in self.direct_sizes.items() if use]             # an untimed size: as its next larger timed one
        return bool(won) and int(numel) <= max(won)

    def enable_direct_exchange(self, device, max_floats=1 << 21):
        """Use `DirectExchange` for the step's collectives (all ranks must call this; collective).  -> the exchange."""
        self.direct = DirectExchange(self.rank, self.world, self.group, max_floats, device)
        return self.direct

    def _all_ok(self, ok, device):
        """Every rank's verdict AND-ed through the process group (the one path that is known to work)."""
        t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float32, device=device if dist.get_backend(self.group) == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return float(t.item()) > 0

    def try_direct_exchange(self, device, max_floats=1 << 21, selftest_timeout_ms=500):
        """Set the one-shot exchange up IF it works here, on every rank: hipIpc mapping of all mailboxes, then a timed
        self-test (a known sum with a short wait budget, the error word checked).  Any rank's failure -- an exception, a
        wrong sum, a wait that ran out -- makes every rank fall back to the process group.  -> the exchange or None."""
        ex, ok, why = None, True, ''
        try:
            ex = DirectExchange(self.rank, self.world, self.group, max_floats, device)
        except Exception as exc:                                            # noqa: BLE001
            ok, why = False, 'setup: %r' % (exc,)
        if not self._all_ok(ok, device):
            if ex is not None:
                ex.close()
            self.collective_report = {'direct': 'unavailable', 'why': why or 'another rank failed to map the mailboxes'}
            return None
        try:
            ex.set_timeout_ms(selftest_timeout_ms)
            n = min(max_floats, 70001)
            x = torch.full((n,), float(self.rank + 1), dtype=torch.float32, device=device)
            for _ in range(3):                                              # both mailbox halves
                x.fill_(float(self.rank + 1))
                ex.all_reduce(x)
            torch.cuda.synchronize(device)
            ok = ex.error_word() == 0 and bool((x == float(self.world * (self.world + 1) // 2)).all())
            why = '' if ok else 'self-test: wrong sum or a wait ran out (error word %#x)' % ex.error_word()
            # production budget: ranks reach the first exchanges of a step behind host-side work of data-dependent length
            # (plan builds take seconds); a healthy run must not record a timeout there -- RCCL / gloo would simply wait
            ex.set_timeout_ms(PEER_WAIT_MS)
        except Exception as exc:                                            # noqa: BLE001
            ok, why = False, 'self-test: %r' % (exc,)
        if not self._all_ok(ok, device):
            ex.close()
            self.collective_report = {'direct': 'failed its self-test', 'why': why or 'on another rank'}
            return None
        self.direct = ex
        return ex

    def _time_collective_us(self, fn, device, reps):
        """us per call of `fn` the way the step issues it -- `reps` calls captured into ONE hipGraph and replayed, HIP events on
        the replay stream -- or, if this build cannot capture the call on some rank, `reps` eager calls, host-synchronised
        (every rank takes the same branch: the verdict travels through the process group).  -> (us, how)."""
        import time
        graph, ok = None, True
        try:
            if dist.get_backend(self.group) != 'nccl':                      # (a host-side backend cannot be captured; its step
                raise RuntimeError('eager backend')                         #  runs eagerly as well: bench.py `launch`)
            side = torch.cuda.Stream(device)
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):
                for _ in range(3):
                    fn()
            torch.cuda.current_stream(device).wait_stream(side)
            torch.cuda.synchronize(device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode='thread_local'):
                for _ in range(reps):
                    fn()
        except Exception:                                                   # noqa: BLE001
            ok, graph = False, None
            torch.cuda.synchronize(device)
        if self._all_ok(ok, device):
            graph.replay()
            torch.cuda.synchronize(device)
            dist.barrier(group=self.group)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(3):
                graph.replay()
            b.record()
            torch.cuda.synchronize(device)
            return a.elapsed_time(b) * 1e3 / (3 * reps), 'hipGraph replay'
        for _ in range(3):
            fn()
        torch.cuda.synchronize(device)
        dist.barrier(group=self.group)
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize(device)
        return (time.perf_counter() - t0) / reps * 1e6, 'eager calls, host-synchronised'

    def choose_collective(self, sizes, device, reps=20):
        """Time the process group's all-reduce and the one-shot exchange for every message size of the step (floats) the way
        the step uses them -- inside a captured hipGraph (`_time_collective_us`), the MAX over the ranks -- and keep the
        faster per size (the same decision on every rank).  Call before the step is captured.  -> the report (also
        `self.collective_report`)."""
        assert self.direct is not None
        rep = {}
        use = {}
        how = set()
        on_dev = dist.get_backend(self.group) == 'nccl'
        for n in sorted(set(int(s_) for s_ in sizes)):
            if n <= 0 or n > self.direct.max_floats:
                continue
            x = torch.zeros(n, dtype=torch.float32, device=device)
            t = {}
            for name in ('group', 'direct'):
                fn = (lambda: dist.all_reduce(x, op=dist.ReduceOp.SUM, group=self.group)) if name == 'group' else \
                    (lambda: self.direct.all_reduce(x))
                t[name], h = self._time_collective_us(fn, device, reps)
                how.add(h)
            tt = torch.tensor([t['group'], t['direct']], dtype=torch.float32, device=device if on_dev else 'cpu')
            dist.all_reduce(tt, op=dist.ReduceOp.MAX, group=self.group)
            g_us, d_us = [float(v) for v in tt.tolist()]
            use[n] = d_us < g_us
            rep[str(n)] = {'group_us': round(g_us, 1), 'direct_us': round(d_us, 1), 'chosen': 'direct' if use[n] else 'group'}
        self.direct.check()
        self.direct_sizes = use
        self.collective_report = {'per_size': rep, 'group_backend': dist.get_backend(self.group),
                                  'how': ' / '.join(sorted(how)) + ', max over ranks'}
        return self.collective_report

    @property
    def collective(self):
        group = dist.get_backend(self.group) if dist.is_initialized() else 'none'
        if self.direct is None:
            return group
        if self.direct_sizes is None or all(self.direct_sizes.values()):
            return 'direct'
        if not any(self.direct_sizes.values()):
            return group
        return 'direct<=%d floats, %s above' % (max(n for n, u in self.direct_sizes.items() if u), group)

    @property
    def loss_weight(self):
        return float(self.n_train_local) / float(max(1, self.n_train_total))


class PeerTimeout(RuntimeError):
    """An exchange of `DirectExchange` ran out of its wait budget: fatal for the process group."""


class DirectExchange(object):
    """One-shot all-reduce over peer-mapped mailboxes (include/tipk.h section 8, tip_amd/csrc/tipk_peer.hip): every rank
    writes its buffer into its slot of all ranks' mailboxes -- w - 1 remote streams over w - 1 different xGMI links at
    once -- posts flags, waits for the others' flags and adds the slots in rank order.  One kernel per rank and call
    instead of the 2 (w - 1) dependent steps of a ring; results are identical on every rank and from run to run.
    The mailboxes are exchanged as hipIpc handles through the process group (any backend) once, at construction.
    It is validated with ranks SHARING one GPU (tests/test_gpu_direct_exchange.py: 2, 4 and 8 ranks, a peer that skips a
    call); over xGMI it has never run.  `RelationShard.try_direct_exchange` therefore sets it up only if every rank can map
    every mailbox and a timed self-test gives the right sums, `choose_collective` then times it against the process
    group per message size and keeps the faster (bench.py does both for N > 1); the flag wait is bounded, and a wait that
    ran out is reported by `check()`."""

    def __init__(self, rank, world, group, max_floats, device):
        """Collective over `group`.  A rank whose mailbox cannot be allocated or mapped does NOT leave the others waiting
        in the handle exchange: failures travel with the handles, and every rank raises together."""
        import ctypes as C
        from ._lib import lib, check
        L = lib()
        self.rank, self.world, self.max_floats, self.device = int(rank), int(world), int(max_floats), torch.device(device)
        self.own, self.opened = None, []
        nbytes = int(L.tipk_peer_mailbox_bytes(self.world, self.max_floats))
        if nbytes <= 0:
            raise ValueError('direct exchange: unsupported world size %d' % world)
        err = None
        handle = C.create_string_buffer(64)
        with torch.cuda.device(self.device):
            try:
                own = C.c_void_p()
                check(L.tipk_peer_alloc(nbytes, C.byref(own)), 'tipk_peer_alloc')
                self.own = own
                check(L.tipk_ipc_get_handle(self.own, handle), 'tipk_ipc_get_handle')
            except Exception as exc:                                        # noqa: BLE001
                err = repr(exc)
            handles = [None] * self.world
            mine = (handle.raw if err is None else None, err)
            if self.world > 1:
                dist.all_gather_object(handles, mine, group=group)
            else:
                handles = [mine]
            bad = [(r, h[1]) for r, h in enumerate(handles) if h[0] is None]
            if not bad:
                self.ptrs = (C.c_void_p * self.world)()
                try:
                    for r in range(self.world):
                        if r == self.rank:
                            self.ptrs[r] = self.own
                        else:
                            p = C.c_void_p()
                            check(L.tipk_ipc_open(handles[r][0], C.byref(p)), 'tipk_ipc_open')
                            self.ptrs[r] = p
                            self.opened.append(p)
                except Exception as exc:                                    # noqa: BLE001
                    err = repr(exc)
                oks = [None] * self.world
                if self.world > 1:
                    dist.all_gather_object(oks, err, group=group)           # (also: every mailbox is mapped everywhere
                else:                                                       #  before the first exchange)
                    oks = [err]
                bad = [(r, e) for r, e in enumerate(oks) if e is not None]
        if bad:
            self.close()
            raise RuntimeError('direct exchange unavailable: ' + '; '.join('rank %d: %s' % b for b in bad))

    def takes(self, flat):
        return (flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous() and 0 < flat.numel() <= self.max_floats
                and flat.device == self.device)

    def all_reduce(self, flat):
        from ._lib import lib, check, ptr, stream_ptr
        check(lib().tipk_peer_allreduce(ptr(flat), flat.numel(), self.ptrs, self.rank, self.world, self.max_floats,
                                        stream_ptr(flat.device)), 'tipk_peer_allreduce')
        return flat

    def set_timeout_ms(self, ms):
        """Wall-clock budget of the flag wait of later exchanges (PROCESS-WIDE in libtipk: include/tipk.h section 8; the
        library's own default is 2 000 ms, `try_direct_exchange` leaves PEER_WAIT_MS behind)."""
        from ._lib import lib, check
        check(lib().tipk_peer_set_timeout_ms(int(ms)), 'tipk_peer_set_timeout_ms')

    def error_word(self):
        """0, or the record of the first flag wait that ran out of its budget (a synchronous 8-byte copy)."""
        import ctypes as C
        from ._lib import lib, check
        if self.own is None:
            return 0
        w = C.c_uint64(0)
        with torch.cuda.device(self.device):
            check(lib().tipk_peer_status(self.own, self.world, self.max_floats, C.byref(w)), 'tipk_peer_status')
        return int(w.value)

    def check(self):
        """Raise PeerTimeout if an exchange gave up waiting for a peer (its result was garbage): call where the host
        synchronises anyway -- after a step's loss has been read, after a timed region.  The process group is dead
        after that: the caller exits non-zero (it must NOT re-exec itself: the process has touched the GPU)."""
        w = self.error_word()
        if w:
            raise PeerTimeout('rank %d: exchange %d gave up waiting for rank %d (chunk %d): a peer died, skipped a call or '
                              'took another collective' % (self.rank, w >> 32, ((w >> 16) & 0xffff) - 1, w & 0xffff))

    def close(self):
        from ._lib import lib
        if getattr(self, 'own', None) is None and not getattr(self, 'opened', None):
            return
        torch.cuda.synchronize(self.device)
        for p in self.opened:
            lib().tipk_ipc_close(p)
        self.opened = []
        if self.own is not None:
            lib().tipk_peer_free(self.own)
            self.own = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:                                   # interpreter shutdown: the driver reclaims the mappings
            pass


def make_shard(range_list, rank, world, group=None):
    """Partition by the edge counts of `range_list` [R, 2] -> this rank's `RelationShard`."""
    rg = torch.as_tensor(range_list).to(torch.int64).cpu()
    parts = partition_relations((rg[:, 1] - rg[:, 0]).tolist(), world)
    return RelationShard(parts[rank], rank, world, group, n_relations=rg.shape[0])


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


def _local_ranges(range_list, rel_ids):
    rg = torch.as_tensor(range_list).to(torch.int64).cpu()
    sizes = (rg[:, 1] - rg[:, 0])[torch.as_tensor(rel_ids, dtype=torch.int64)] if len(rel_ids) else \
        torch.zeros(0, dtype=torch.int64)
    end = torch.cumsum(sizes, 0)
    return torch.stack([end - sizes, end], dim=1)


def shard_data_dict(dd, shard):
    """The `data_dict` of ONE rank: its relations' train and test edges with local
    relation ids and ranges, `n_dd_et` = local relation count; everything else (features, P-P, P->D)
    is shared.  Records the global in-degree and triple count on `shard`."""
    ids = shard.rel_ids.tolist()
    out = {k: v for k, v in dd.items() if not k.startswith('dd_')}
    n = dd['n_drug']
    shard.in_degree = torch.bincount(dd['dd_train_idx'][1].to(torch.int64).cpu(), minlength=n)
    shard.n_train_total = int(dd['dd_train_idx'].shape[1])
    for split in ('train', 'test'):
        idx_k, rg_k = 'dd_%s_idx' % split, 'dd_%s_range' % split
        if idx_k not in dd:
            continue
        ei, rel = shard_edges(dd[idx_k], dd[rg_k], ids)
        out[idx_k], out['dd_%s_et' % split], out[rg_k] = ei.contiguous(), rel, _local_ranges(dd[rg_k], ids)
        # position of a relation's block in the WHOLE list minus its position in the local one: the sampler's Philox
        # counter runs over global positions, so the negatives do not depend on the number of ranks
        rg_all = torch.as_tensor(dd[rg_k]).to(torch.int64).cpu()
        out['dd_%s_pos_offset' % split] = (rg_all[ids, 0] - out[rg_k][:, 0]) if len(ids) else torch.zeros(0, dtype=torch.int64)
    shard.n_train_local = int(out['dd_train_idx'].shape[1])
    out['n_dd_et'] = len(ids)
    out['dd_rel_ids'] = shard.rel_ids.clone()
    if 'dd_edge_index' in dd:
        out['dd_edge_index'] = [dd['dd_edge_index'][r] for r in ids]
    return out


def shard_state_dict(full_sd, shard):
    """Rows of the per-relation parameters that belong to this rank (everything else unchanged)."""
    ids = shard.rel_ids
    out = {}
    for k, v in full_sd.items():
        out[k] = v.index_select(0, ids.to(v.device)).clone() if k in LOCAL_ROWS or ('encoder.' + k) in LOCAL_ROWS \
            else v.clone()
    return out


def gather_state_dict(model, shard):
    """Full (unsharded) state_dict on every rank: the shard-local rows are all-gathered once -- for
    checkpoints (`torch.save`, tip.py:36); never needed during training."""
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    if shard.world == 1 and not dist.is_initialized():
        gathered = [(shard.rel_ids, {k: sd[k].cpu() for k in LOCAL_ROWS if k in sd})]
    else:
        gathered = [None] * shard.world
        dist.all_gather_object(gathered, (shard.rel_ids.cpu(), {k: sd[k].cpu() for k in LOCAL_ROWS if k in sd}),
                               group=shard.group)
    full = {k: v.cpu().clone() for k, v in sd.items() if k not in LOCAL_ROWS}
    for k in LOCAL_ROWS:
        if k not in sd:
            continue
        rows = torch.zeros((shard.n_relations,) + tuple(sd[k].shape[1:]), dtype=sd[k].dtype)
        for ids, part in gathered:
            rows[ids] = part[k]
        full[k] = rows
    return full


def attach_shard(encoder, shard):
    """Make both R-GCN layers of an `FMEncoder` (built for the LOCAL relation count over the local edge
    tensors of `shard_data_dict`) reduce their partial sums over the shard's group."""
    for layer in (encoder.rgcn1, encoder.rgcn2):
        assert layer.num_relations == int(shard.rel_ids.numel()), 'build the encoder for the local relation count'
        layer.shard = shard
        layer._cache.key = None                 # plans of an earlier (unsharded) call are stale
    return shard


class _AllReduceSum(torch.autograd.Function):
    """total = sum over ranks of a (scalar or small) tensor; d total / d local = 1 on every rank."""

    @staticmethod
    def forward(ctx, t, shard):
        out = t.detach().clone()
        shard.all_reduce(out)
        return out

    @staticmethod
    def backward(ctx, g):
        return g, None


class _SumGradOverRanks(torch.autograd.Function):
    """Identity on a tensor that is identical on every rank and consumed by shard-local work: its
    gradient is the SUM of the ranks' partial gradients (the d z all-reduce of SURVEY 8(e))."""

    @staticmethod
    def forward(ctx, t, shard):
        ctx.shard = shard
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().clone()
        ctx.shard.all_reduce(g)
        return g, None


def all_reduce_sum(t, shard):
    return _AllReduceSum.apply(t, shard)


def sum_grad_over_ranks(t, shard):
    return _SumGradOverRanks.apply(t, shard)
