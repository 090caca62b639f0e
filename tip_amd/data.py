"""Graph ingest for the TIP hot path: the `data_dict.pkl` schema without the 476 MB pickle.

The reference trains from `data/data_dict.pkl`, produced by `prepare.py:10-47` (schema: SURVEY.md
section 8(a) row A0).  Only this repository travels to the GPU box, so the BioSNAP graph ships as a
6 MB compact binary (`tip_amd/data/biosnap_v1.npz`, made by `tools/make_biosnap_blob.py` from the
reference's `data/sym_adj/**`) and `build_data_dict` replays the reference's split on it with a
seeded legacy numpy generator.  `synthetic_data_dict` builds BASELINE.json's config 5 shape.

A pickle written by the reference's own `prepare.py` loads unchanged through `TIP(data_path=...)`.
"""
import os

import numpy as np
import torch

from .utils import process_edges, process_prot_edge, sparse_id, to_bidirection, get_range_list

BIOSNAP_BLOB = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data', 'biosnap_v1.npz')
MONO_BLOB = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data', 'biosnap_mono_v1.npz')


def mono_drug_features(blob_path=MONO_BLOB):
    """(d_feat, d_norm) with the REAL drug features the reference prepared but never switched on
    (`prepare.py:21` "TODO: add drug feature"):

    d_feat  sparse fp32 [n_drug, n_drug + n_mono] = [ I | mono side-effect indicators ], built exactly as
            `data/utils.py:117-132` (mono=True) builds it: identity entries first, then the mono entries
            with their column offset by n_drug, all values 1;
    d_norm  fp32 [n_drug] = features per drug (1 + its mono side effects).  The reference defines no
            working normaliser for this case (`prepare.py:25` sizes `d_norm` by the feature count, which
            cannot divide the 645 rows at `src/layers.py:534`); dividing by the row sum makes
            `x_drug @ embed / d_norm` the MEAN of a drug's feature embeddings."""
    z = np.load(blob_path)
    n_drug, n_mono = int(z['n_drug']), int(z['n_mono'])
    pairs = z['mono_pairs'].astype(np.int64)
    row = np.r_[np.arange(n_drug), pairs[0]]
    col = np.r_[np.arange(n_drug), pairs[1] + n_drug]
    d_feat = torch.sparse_coo_tensor(torch.from_numpy(np.stack([row, col])), torch.ones(row.size),
                                     (n_drug, n_drug + n_mono)).coalesce()
    d_norm = torch.from_numpy(np.bincount(row, minlength=n_drug).astype(np.float32))
    return d_feat, d_norm


class Data(object):
    """Attribute bag with `.to(device)` -- the slice of PyG's `Data` the reference uses
    (`src/layers.py:288,280`): tensors (also inside lists/tuples/dicts) move, the rest stays."""

    @classmethod
    def from_dict(cls, dictionary):
        obj = cls()
        obj.__dict__.update(dictionary)
        return obj

    def to(self, device):
        def mv(v):
            if isinstance(v, torch.Tensor):
                return v.to(device)
            if isinstance(v, (list, tuple)):
                return type(v)(mv(u) for u in v)
            if isinstance(v, dict):
                return {k: mv(u) for k, u in v.items()}
            return v
        for k in list(self.__dict__):
            self.__dict__[k] = mv(self.__dict__[k])
        return self

    def keys(self):
        return list(self.__dict__)


def _dp_tables(dp_pairs, n_drug, n_prot):
    """`dp_edge_index` / `dp_range_list` exactly as `prepare.py:30-44` derives them from the
    (protein-1, drug-1) COO pairs: targets are offset by n_prot, ranges count edges per drug."""
    prot = torch.as_tensor(dp_pairs[0].astype(np.int64))
    drug = torch.as_tensor(dp_pairs[1].astype(np.int64))
    cnt = torch.bincount(drug, minlength=n_drug)
    end = torch.cumsum(cnt, 0)
    dp_range = torch.stack([end - cnt, end], dim=1).to(torch.float32)     # prepare.py:44 torch.Tensor
    return torch.stack([prot, drug + n_prot]), dp_range


def build_data_dict(blob_path=BIOSNAP_BLOB, sp_rate=0.9, seed=1111, min_pairs=None, max_relations=None,
                    mono=False):
    """BioSNAP `data_dict` (tensors on CPU).

    mono         False: identity drug features and unit `d_norm`, as `prepare.py:22-25` ships them;
                 True: the real features of `mono_drug_features` (n_drug_feat = 10 829).
    seed         legacy numpy seed for the Bernoulli split (`np.random.seed(1111)`,
                 `src/layers.py:14`); draw order = `prepare.py:16-17`: all D-D relations, then P-P.
    min_pairs    keep only relations with at least this many undirected pairs (500 -> the 963
                 relations of the paper runs); None keeps the shipped 1097.
    max_relations  keep the first k relations (parity-test slices).
    """
    z = np.load(blob_path)
    n_drug, n_prot = int(z['n_drug']), int(z['n_prot'])
    ptr = z['dd_ptr']
    pairs = torch.from_numpy(z['dd_pairs'].astype(np.int64))
    keep = list(range(len(ptr) - 1))
    if min_pairs is not None:
        keep = [r for r in keep if ptr[r + 1] - ptr[r] >= min_pairs]
    if max_relations is not None:
        keep = keep[:max_relations]
    raw = [pairs[:, ptr[r]:ptr[r + 1]] for r in keep]

    rng = np.random.RandomState(seed)
    d = {}
    (d['dd_train_idx'], d['dd_train_et'], d['dd_train_range'],
     d['dd_test_idx'], d['dd_test_et'], d['dd_test_range']) = process_edges(raw, p=sp_rate, rng=rng)
    pp = torch.from_numpy(z['pp_pairs'].astype(np.int64))
    d['dd_edge_index'] = raw                      # un-split pairs per relation (re-split if sp_rate != 0.9)
    d['pp_train_indices'], d['pp_test_indices'] = process_prot_edge(pp, rng=rng)
    d['dp_edge_index'], d['dp_range_list'] = _dp_tables(z['dp_pairs'], n_drug, n_prot)
    d['d_feat'] = sparse_id(n_drug)                                       # prepare.py:22-23
    d['p_feat'] = sparse_id(n_prot)
    d['n_drug'], d['n_prot'], d['n_dd_et'] = n_drug, n_prot, len(keep)
    d['n_drug_feat'] = n_drug
    d['d_norm'] = torch.ones(n_drug)
    if mono:
        d['d_feat'], d['d_norm'] = mono_drug_features()
        d['n_drug_feat'] = int(d['d_feat'].shape[1])
    d['et_list'] = [int(z['et_list'][r]) for r in keep]
    return d


def sym_adj_to_pairs(indptr, indices, n_nodes):
    """Undirected pair list of a stack of SYMMETRIC adjacency matrices in the layout of the
    reference's `data/sym_adj/drug-sparse-adj/type_*.npz` (scipy CSR), without per-relation Python
    lists: `indptr` int64 [R, n_nodes + 1] (row pointers of every relation, each relative to that
    relation's first entry), `indices` int64 [nnz] (all relations' column ids, concatenated).
    -> (pairs int64 [2, P] with row < col in CSR order, rel_ptr int64 [R + 1]) on the arrays' device:
    what `sp.triu(adj).tocsr().tocoo()` yields per relation (data/utils.py:60,151)."""
    R = indptr.shape[0]
    nnz_r = indptr[:, -1]
    base = torch.cumsum(nnz_r, 0) - nnz_r
    counts = (indptr[:, 1:] - indptr[:, :-1]).reshape(-1)                       # entries per (relation, row)
    row = torch.repeat_interleave(torch.arange(n_nodes, device=indices.device).repeat(R), counts)
    rel = torch.repeat_interleave(torch.arange(R, device=indices.device), nnz_r)
    keep = row < indices
    pairs = torch.stack([row[keep], indices[keep]])
    rel_ptr = torch.zeros(R + 1, dtype=torch.int64, device=indices.device)
    rel_ptr[1:] = torch.cumsum(torch.bincount(rel[keep], minlength=R), 0)
    del base
    return pairs, rel_ptr


def device_process_edges(pairs, rel_ptr, p=0.9, seed=1111):
    """`process_edges` (src/utils.py:35-65) on the GPU: seeded Philox Bernoulli(p) split of the
    concatenated undirected pairs + mirroring, two launches (`tipk_split_flags`, `tipk_split_scatter`),
    no per-relation host work.  pairs: [2, P] device tensor (int64 or uint16 stored as int16 is not
    accepted: int64 / uint16); rel_ptr int64 [R + 1].  Returns the reference's six tensors, on the
    device: train_idx [2,E], train_et, train_range [R,2], test_idx, test_et, test_range."""
    from . import _lib
    from ._lib import check, lib, ptr, stream_ptr, require_device
    require_device(pairs, rel_ptr)
    assert pairs.dim() == 2 and pairs.shape[0] == 2 and rel_ptr.dtype == torch.int64
    if pairs.dtype not in (torch.int64, torch.uint16):
        raise _lib.TipkError('pairs must be int64 or uint16, got %s' % pairs.dtype)
    pairs = pairs.contiguous()
    dev = pairs.device
    P, R = pairs.shape[1], rel_ptr.numel() - 1
    st = stream_ptr(dev)
    take = torch.empty(P, dtype=torch.uint8, device=dev)
    n_train = torch.zeros(R, dtype=torch.int64, device=dev)
    check(lib().tipk_split_flags(ptr(rel_ptr), R, P, float(p), int(seed) & ((1 << 64) - 1), ptr(take), ptr(n_train), st),
          'tipk_split_flags')
    n_all = rel_ptr[1:] - rel_ptr[:-1]
    zero = torch.zeros(1, dtype=torch.int64, device=dev)
    train_ptr = torch.cat([zero, torch.cumsum(2 * n_train, 0)])
    test_ptr = torch.cat([zero, torch.cumsum(2 * (n_all - n_train), 0)])
    e_tr, e_te = int(train_ptr[-1]), int(test_ptr[-1])                       # the only host sync (allocation sizes)
    tr = torch.empty((2, e_tr), dtype=torch.int64, device=dev)
    te = torch.empty((2, e_te), dtype=torch.int64, device=dev)
    tr_et = torch.empty(e_tr, dtype=torch.int64, device=dev)
    te_et = torch.empty(e_te, dtype=torch.int64, device=dev)
    check(lib().tipk_split_scatter(ptr(pairs[0]), ptr(pairs[1]), 8 if pairs.dtype == torch.int64 else 2, ptr(rel_ptr), R,
                                   ptr(take), ptr(train_ptr), ptr(test_ptr), ptr(tr[0]), ptr(tr[1]), ptr(tr_et),
                                   ptr(te[0]), ptr(te[1]), ptr(te_et), st), 'tipk_split_scatter')
    return (tr, tr_et, torch.stack([train_ptr[:-1], train_ptr[1:]], 1),
            te, te_et, torch.stack([test_ptr[:-1], test_ptr[1:]], 1))


def build_data_dict_device(device, blob_path=BIOSNAP_BLOB, sp_rate=0.9, seed=1111, max_relations=None):
    """The BioSNAP `data_dict` built ON THE DEVICE: the blob's pair
    arrays go to HBM once and the D-D and P-P splits + mirroring run there (`device_process_edges`),
    replacing the reference's prepare.py -> 476 MB pickle -> unpickle round trip.  Same schema as
    `build_data_dict`; the split is this build's seeded Philox stream (same distribution as the
    reference's Bernoulli draw, not the same sample)."""
    z = np.load(blob_path)
    n_drug, n_prot = int(z['n_drug']), int(z['n_prot'])
    ptr_np = z['dd_ptr'].astype(np.int64)
    R = len(ptr_np) - 1 if max_relations is None else min(max_relations, len(ptr_np) - 1)
    dev = torch.device(device)
    pairs = torch.from_numpy(z['dd_pairs'][:, :ptr_np[R]].astype(np.int64)).to(dev)
    rel_ptr = torch.from_numpy(ptr_np[:R + 1]).to(dev)
    d = {}
    (d['dd_train_idx'], d['dd_train_et'], d['dd_train_range'],
     d['dd_test_idx'], d['dd_test_et'], d['dd_test_range']) = device_process_edges(pairs, rel_ptr, sp_rate, seed)
    pp = torch.from_numpy(z['pp_pairs'].astype(np.int64)).to(dev)
    pp_ptr = torch.tensor([0, pp.shape[1]], dtype=torch.int64, device=dev)
    tr, _, _, te, _, _ = device_process_edges(pp, pp_ptr, sp_rate, seed + 1)     # data/utils.py:212-229
    d['pp_train_indices'], d['pp_test_indices'] = tr, te
    dp_idx, dp_rg = _dp_tables(z['dp_pairs'], n_drug, n_prot)
    d['dp_edge_index'], d['dp_range_list'] = dp_idx.to(dev), dp_rg.to(dev)
    d['d_feat'], d['p_feat'] = sparse_id(n_drug).to(dev), sparse_id(n_prot).to(dev)
    d['n_drug'], d['n_prot'], d['n_dd_et'], d['n_drug_feat'] = n_drug, n_prot, R, n_drug
    d['d_norm'] = torch.ones(n_drug, device=dev)
    d['et_list'] = [int(t) for t in z['et_list'][:R]]
    return d


def lognormal_sizes(total, n, sigma, rng, minimum=2):
    """n even block sizes ~ lognormal(sigma) rescaled to sum to `total` (BioSNAP's measured skew
    is sigma_ln = 1.17)."""
    w = rng.lognormal(0.0, sigma, n)
    s = np.maximum(minimum, 2 * np.round(w / w.sum() * total / 2).astype(np.int64))
    s[np.argmax(s)] += total - s.sum()
    assert s.sum() == total and (s % 2 == 0).all() and (s > 0).all()
    return s


def synthetic_data_dict(n_drug=10000, n_rel=2000, n_edges=50_000_000, seed=1111, sigma=1.17,
                        with_protein_graph=False, n_prot=2048, pp_edges=16384, dp_edges=4096):
    """Synthetic D-D graph in the `data_dict` schema: `n_edges` directed edges = mirrored
    undirected pairs, relation sizes lognormal, endpoints uniform without self pairs.  The
    default is BASELINE.json config 5 (no P-P / P->D stages: X is fed directly); with
    `with_protein_graph` a small random protein side is attached so the full encoder runs."""
    rng = np.random.RandomState(seed)
    sizes = lognormal_sizes(n_edges, n_rel, sigma, rng)
    half = sizes // 2
    tot = int(half.sum())
    u = torch.from_numpy(rng.randint(0, n_drug, tot).astype(np.int64))
    v = torch.from_numpy(rng.randint(0, n_drug - 1, tot).astype(np.int64))
    v = v + (v >= u).to(torch.int64)                                      # no self pairs
    off = np.r_[0, np.cumsum(half)]
    blocks = [to_bidirection(torch.stack([u[off[r]:off[r + 1]], v[off[r]:off[r + 1]]]))
              for r in range(n_rel)]
    d = {'dd_train_idx': torch.cat(blocks, dim=1),
         'dd_train_et': torch.repeat_interleave(torch.arange(n_rel), torch.from_numpy(sizes)),
         'dd_train_range': get_range_list(blocks),
         'n_drug': n_drug, 'n_dd_et': n_rel, 'n_drug_feat': n_drug,
         'd_norm': torch.ones(n_drug)}
    if with_protein_graph:
        a = torch.from_numpy(rng.randint(0, n_prot, (2, pp_edges)).astype(np.int64))
        a = a[:, a[0] != a[1]]
        d['pp_train_indices'] = to_bidirection(a)
        pr = rng.randint(0, n_prot, dp_edges)
        dr = np.sort(rng.randint(0, n_drug, dp_edges))
        d['dp_edge_index'], d['dp_range_list'] = _dp_tables(np.stack([pr, dr]), n_drug, n_prot)
        d['d_feat'], d['p_feat'], d['n_prot'] = sparse_id(n_drug), sparse_id(n_prot), n_prot
        d['dd_test_idx'], d['dd_test_et'], d['dd_test_range'] = (
            d['dd_train_idx'][:, :0], d['dd_train_et'][:0], torch.zeros((n_rel, 2), dtype=torch.long))
    return d
