"""TEST INFRASTRUCTURE -- bit-exact numpy specification of the DEVICE train/test split
(`tipk_split_flags` / `tipk_split_scatter`, include/tipk.h section 7).

This is not a restatement of the reference's draw: `process_edges` (src/utils.py:35-65) takes one
`np.random.binomial(1, p, E_r)` per relation from numpy's global Mersenne state, which no device RNG
reproduces (`tip_amd.utils.process_edges` replays that one on the host, bit for bit, for the bundled
graph).  What IS the reference's is everything after the draw -- kept pairs in list order, mirrored
halves appended per relation (`to_bidirection`, :17-23), edge types, ranges -- and this spec applies
exactly that to the Philox flags:

    pair i (index in the concatenated pair list) is a TRAINING pair iff
        Philox4x32-10(counter = (i & 0xffffffff, i >> 32, 0, 0x53504C54), key = (seed lo, seed hi)).x0
            < floor(p * 2^32)                                   (p = 1: every pair)
"""
import numpy as np

from .philox_sampler import philox4x32_10

TAG = 0x53504C54            # 'SPLT'


def split_flags_spec(n_pairs, p, seed):
    i = np.arange(n_pairs, dtype=np.uint64)
    x0, _, _, _ = philox4x32_10(i & np.uint64(0xFFFFFFFF), i >> np.uint64(32), np.zeros(n_pairs), TAG,
                                seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    if p * 4294967296.0 >= 4294967296.0:
        return np.ones(n_pairs, dtype=bool)
    return x0.astype(np.uint64) < np.uint64(int(p * 4294967296.0))


def process_edges_spec(pairs, rel_ptr, p, seed):
    """pairs int64 [2, P]; rel_ptr [R+1] -> the six arrays of `process_edges` (int64):
    train_idx [2,E], train_et [E], train_range [R,2], test_idx, test_et, test_range."""
    pairs = np.asarray(pairs, dtype=np.int64)
    rel_ptr = np.asarray(rel_ptr, dtype=np.int64)
    take = split_flags_spec(pairs.shape[1], p, seed)
    out = []
    for keep in (take, ~take):
        blocks, ets, rg, pos = [], [], [], 0
        for r in range(rel_ptr.size - 1):
            a, b = rel_ptr[r], rel_ptr[r + 1]
            sel = pairs[:, a:b][:, keep[a:b]]
            blk = np.concatenate([sel, sel[::-1]], axis=1)               # to_bidirection: [E | E mirrored]
            blocks.append(blk)
            ets.append(np.full(blk.shape[1], r, dtype=np.int64))
            rg.append((pos, pos + blk.shape[1]))
            pos += blk.shape[1]
        out += [np.concatenate(blocks, axis=1) if blocks else np.zeros((2, 0), np.int64),
                np.concatenate(ets) if ets else np.zeros(0, np.int64), np.asarray(rg, dtype=np.int64).reshape(-1, 2)]
    return out
