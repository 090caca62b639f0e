"""Typed negative sampling on the GPU (mirror of the reference's `src/neg_sampling.py:5-26`).

Same call signature and result contract: for every relation block `[start, end)` of
`range_list`, as many pairs as the block has positives, drawn uniformly with replacement from
`num_nodes ** 2` (self pairs and repeats allowed, exactly like `np.random.choice`), none of them
equal to a positive pair of the SAME relation; result int64 `[2, E]` on the positives' device.

Differences, on purpose:
  * the reference makes one device->host copy and one numpy sort per relation per epoch
    (1 097 round trips); here the sorted positive keys are built once per positive tensor and one
    kernel draws all relations (`tipk_typed_negative_sampling`);
  * the reference's resample loop re-indexes `rest` into the wrong array (:13-16), so a few
    sampled pairs ARE positives (879 of 51 546 in the densest relation); this sampler implements
    the intended rejection, so none is;
  * randomness is counter-based Philox4x32-10 keyed by `(seed, call counter)` instead of numpy's
    global Mersenne state: `manual_seed` makes a run reproducible, sample-level equality with
    numpy is impossible by construction (tests check the distribution instead).
"""
import torch

from . import ops

_MASK = (1 << 64) - 1
_state = {'seed': 1111}
_counters = {}                 # device -> int64 [3] tensor {position, seed, ticket}: the stream's state ON THE DEVICE
_key_cache = {}


def manual_seed(seed):
    """Reset the sampler stream (the analogue of `np.random.seed`, src/layers.py:14).  Call n of the
    stream uses the Philox key `call_key(seed, n)`; BOTH n and the seed live in device memory, so a
    captured hipGraph (tip_amd.train.GraphedTrainStep) draws fresh negatives on every replay and
    re-seeding after capture takes effect in the replays."""
    _state['seed'] = int(seed) & _MASK
    for c in _counters.values():
        c.copy_(_state_words(_state['seed']))


def _state_words(seed):
    """{position 0, seed, ticket 0} as int64 (the seed's bit pattern).  The ticket word lets the sampling launch
    advance the position itself (include/tipk.h section 5, advance != 0)."""
    s = seed - (1 << 64) if seed >= (1 << 63) else seed
    return torch.tensor([0, s, 0], dtype=torch.int64)


def _counter(device):
    c = _counters.get(device)
    if c is None:
        c = _counters[device] = _state_words(_state['seed']).to(device)
    return c


def stream_position(device):
    """Number of draws made on `device` since the last `manual_seed` (host copy: synchronises)."""
    return int(_counter(torch.device(device))[0])


def call_key(seed, n):
    """Philox key of call n (host mirror of the device function in tipk_negsample.hip)."""
    return _mix(seed & _MASK, n)


def _mix(seed, n):
    """splitmix64 of (seed + n * golden): the 64-bit Philox key of call n."""
    z = (seed + (n + 1) * 0x9E3779B97F4A7C15) & _MASK
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK
    return z ^ (z >> 31)


def relation_ptr(range_list, n_edges):
    """[R+1] int64 offsets from the reference's [R,2] (start,end) table; blocks must tile [0,E)."""
    rg = torch.as_tensor(range_list).to(torch.int64).cpu()
    if rg.numel() == 0:
        return torch.zeros(1, dtype=torch.int64)
    ok = int(rg[0, 0]) == 0 and bool((rg[1:, 0] == rg[:-1, 1]).all()) and int(rg[-1, 1]) == n_edges
    if not ok:
        raise ValueError('range_list must be consecutive blocks covering all %d positions' % n_edges)
    return torch.cat([rg[:1, 0], rg[:, 1]])


def sorted_positive_keys(pos_edge_index, num_nodes, rel_ptr):
    """key = u * num_nodes + v, sorted ascending inside each relation block (one-off)."""
    dev = pos_edge_index.device
    key = pos_edge_index[0].to(torch.int64) * num_nodes + pos_edge_index[1].to(torch.int64)
    sizes = (rel_ptr[1:] - rel_ptr[:-1]).to(dev)
    rel = torch.repeat_interleave(torch.arange(sizes.numel(), device=dev), sizes)
    # one global sort of (relation, key): relation-major order keeps the blocks in place
    order = torch.sort(rel * (num_nodes * num_nodes) + key).indices
    return key[order].contiguous()


def sampler_units(rel_ptr, n_wg, build_cost=0.15, fixed_cost=2048):
    """Deal of the bitmap sampler's work to n_wg workgroups (include/tipk.h section 5): units (relation, first position,
    end position) that tile the positions; a relation above 3/4 of the mean load per workgroup is cut into equal units
    (each of them builds the relation's bitmap: cost build_cost per positive of the relation).  BioSNAP: the largest
    relation has 51 466 positions, the mean load of 256 workgroups is 32 525 -- whole relations leave the slowest
    workgroup 1.58 x the mean.  -> (wg_unit_ptr int32 [n_wg + 1], wg_units int32 [n_units, 3]); deterministic."""
    from .plan import assign_relations
    rel_ptr = [int(x) for x in torch.as_tensor(rel_ptr).tolist()]
    total = rel_ptr[-1]
    cap = max(4096, int(0.75 * total / max(1, n_wg)))
    units, cost = [], []
    for r in range(len(rel_ptr) - 1):
        a, b = rel_ptr[r], rel_ptr[r + 1]
        if b == a:
            continue
        parts = -(-(b - a) // cap)
        for i in range(parts):
            lo, hi = a + (b - a) * i // parts, a + (b - a) * (i + 1) // parts
            units.append((r, lo, hi))
            cost.append(hi - lo + int(build_cost * (b - a)))
    if not units:
        return torch.zeros(n_wg + 1, dtype=torch.int32), torch.zeros((0, 3), dtype=torch.int32)
    assert total < 2 ** 31
    ptr, order = assign_relations(cost, n_wg, fixed_cost=fixed_cost)
    return ptr, torch.tensor([units[i] for i in order.tolist()], dtype=torch.int32).reshape(-1, 3)


def _cached_keys(pos_edge_index, num_nodes, range_list, range_ident=None):
    if range_ident is None:
        range_ident = (range_list.data_ptr(), tuple(range_list.shape)) if torch.is_tensor(range_list) \
            else tuple(map(tuple, range_list))
    ident = (pos_edge_index.data_ptr(), tuple(pos_edge_index.shape), pos_edge_index._version,
             str(pos_edge_index.device), int(num_nodes), range_ident)
    hit = _key_cache.get(ident)
    if hit is None:
        rel_ptr = relation_ptr(range_list, pos_edge_index.shape[1])
        keys = sorted_positive_keys(pos_edge_index, num_nodes, rel_ptr)
        dev = pos_edge_index.device
        n_wg = torch.cuda.get_device_properties(dev).multi_processor_count if dev.type == 'cuda' else 256
        n_wg *= max(1, int(ops.lib().tipk_negsample_wgs_per_cu(int(num_nodes))))       # (two 512-thread workgroups per CU where two bitmaps fit)
        wg_ptr, wg_units = sampler_units(rel_ptr, n_wg)     # (cost-model constants swept in round 6: 0.15 ... 2.0 -> 52.4 ... 53.6 us)
        # the same keys as 32-bit words (n^2 < 2^31): what the bitmap route sets its bits from -- half the bytes per step
        keys32 = keys.to(torch.int32).contiguous() if int(num_nodes) ** 2 < 2 ** 31 and dev.type == 'cuda' else None
        hit = (keys, rel_ptr.to(dev), rel_ptr.numel() - 1, (wg_ptr.to(dev), wg_units.to(dev)), (pos_edge_index, keys32))
        if len(_key_cache) > 8:
            _key_cache.clear()
        _key_cache[ident] = hit
    return hit


def typed_negative_sampling(pos_edge_index, num_nodes, range_list, seed=None, _range_ident=None, pos_offset=None, packed=False):
    """Drop-in for `src/neg_sampling.py:22-26`.  `seed` (optional) pins the Philox key of this
    call; by default consecutive calls use consecutive keys of the `manual_seed` stream.
    pos_offset (extension, int64 [R]): what to add to a position of relation r to get its number in the WHOLE triple
    list of a relation-sharded run (tip_amd/dist.py) -- the rank then draws exactly the unsharded run's negatives.
    packed (extension): return the SAME pairs as int32 [E] words u | v << 16 (num_nodes <= 65535) -- the form the fused
    objective reads (tip_amd/ops.py `distmult_loss`); `ops.unpack_pairs` gives the int64 [2, E] tensor back."""
    num_nodes = int(num_nodes)
    keys, rel_ptr, n_rel, wg, (_, keys32) = _cached_keys(pos_edge_index, num_nodes, range_list, _range_ident)
    if pos_offset is not None:
        pos_offset = pos_offset.to(pos_edge_index.device, torch.int64).contiguous()
        assert pos_offset.numel() == n_rel
    if seed is not None:                                     # explicit Philox key for this call
        return ops.typed_negative_sampling_device(keys, rel_ptr, n_rel, num_nodes, int(seed) & _MASK,
                                                  pos_edge_index.shape[1], dtype=torch.int64, wg=wg, pos_offset=pos_offset, packed=packed,
                                                  keys32=keys32)
    return ops.typed_negative_sampling_device(keys, rel_ptr, n_rel, num_nodes, _state['seed'],
                                              pos_edge_index.shape[1], dtype=torch.int64,
                                              call_counter=_counter(pos_edge_index.device), wg=wg, pos_offset=pos_offset, packed=packed,
                                              keys32=keys32)


def negative_sampling(pos_edge_index, num_nodes, seed=None):
    """Single-relation form (`src/neg_sampling.py:5-19`)."""
    rg = torch.tensor([[0, pos_edge_index.shape[1]]])
    return typed_negative_sampling(pos_edge_index, num_nodes, rg, seed=seed, _range_ident='single')
