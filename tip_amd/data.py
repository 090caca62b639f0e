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
