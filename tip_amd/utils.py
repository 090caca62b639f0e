"""Host-side helpers of the TIP hot path: edge bookkeeping and per-relation metrics.

Mirrors the public names of the reference's `src/utils.py` that the hot path uses
(`process_edges` :35-65, `to_bidirection` :17-23, `remove_bidirection` :7-14, `get_range_list`
:26-32, `sparse_id` :68-75, `auprc_auroc_ap` :86-93) with the same argument meaning.  Everything
here is index arithmetic on the host (one-off per data set); the per-epoch device work lives in
`tip_amd/csrc`.
"""
import numpy as np
import torch

__all__ = ['remove_bidirection', 'to_bidirection', 'get_range_list', 'process_edges',
           'process_prot_edge', 'sparse_id', 'auprc_auroc_ap', 'auprc_auroc_ap_by_range']


def _legacy_rng(rng):
    """The reference draws from numpy's *global* legacy generator; callers that want a
    reproducible split pass a `np.random.RandomState` instead (same bit stream for a seed)."""
    return np.random if rng is None else rng


def remove_bidirection(edge_index, edge_type=None):
    """Keep the edges with source id > target id (reference `src/utils.py:7-14`)."""
    sel = torch.nonzero(edge_index[0] > edge_index[1]).view(-1)
    if edge_type is None:
        return edge_index[:, sel]
    return edge_index[:, sel], edge_type[sel]


def to_bidirection(edge_index, edge_type=None):
    """`[E | E mirrored]` (reference `src/utils.py:17-23`): the mirrored copy is appended, so a
    relation's edge block is always `[u,v half | v,u half]`."""
    both = torch.cat([edge_index, edge_index.flip(0)], dim=1)
    if edge_type is None:
        return both
    return both, torch.cat([edge_type, edge_type])


def get_range_list(edge_list):
    """[R, 2] (start, end) offsets of consecutive blocks (reference `src/utils.py:26-32`)."""
    sizes = torch.tensor([int(e.shape[1]) for e in edge_list], dtype=torch.long)
    end = torch.cumsum(sizes, 0)
    return torch.stack([end - sizes, end], dim=1)


def process_edges(raw_edge_list, p=0.9, rng=None):
    """Bernoulli(p) train/test split of every relation's undirected pairs, then mirroring.

    Same draw order as reference `src/utils.py:35-65` (one `binomial(1, p, E_r)` per relation, in
    list order), so with the same generator state it returns the same six tensors:
    train_idx [2,E], train_et [E], train_range [R,2], test_idx, test_et, test_range.
    """
    gen = _legacy_rng(rng)
    train, test = [], []
    for pairs in raw_edge_list:
        take = gen.binomial(1, p, pairs.shape[1]).astype(bool)
        train.append(to_bidirection(pairs[:, torch.from_numpy(np.flatnonzero(take))]))
        test.append(to_bidirection(pairs[:, torch.from_numpy(np.flatnonzero(~take))]))

    def pack(blocks):
        sizes = torch.tensor([b.shape[1] for b in blocks], dtype=torch.long)
        et = torch.repeat_interleave(torch.arange(len(blocks), dtype=torch.long), sizes)
        return torch.cat(blocks, dim=1), et, get_range_list(blocks)

    tr_idx, tr_et, tr_rg = pack(train)
    te_idx, te_et, te_rg = pack(test)
    return tr_idx, tr_et, tr_rg, te_idx, te_et, te_rg


def process_prot_edge(pp_pairs, p=0.9, rng=None):
    """Split of the P-P graph (reference `data/utils.py:212-229`).  `pp_pairs` is the
    de-duplicated [2, Q] tensor that `remove_bidirection([col; row])` yields there; one
    `binomial(1, p, Q)` draw, train and test halves are mirrored."""
    gen = _legacy_rng(rng)
    take = gen.binomial(1, p, pp_pairs.shape[1]).astype(bool)
    tr = to_bidirection(pp_pairs[:, torch.from_numpy(np.flatnonzero(take))])
    te = to_bidirection(pp_pairs[:, torch.from_numpy(np.flatnonzero(~take))])
    return tr, te


def sparse_id(n):
    """n x n sparse COO identity, fp32 (reference `src/utils.py:68-75`)."""
    i = torch.arange(n, dtype=torch.long)
    return torch.sparse_coo_tensor(torch.stack([i, i]), torch.ones(n), (n, n)).coalesce()


# ---------------------------------------------------------------------------------------------
# metrics (reference `src/utils.py:86-93` calls sklearn; these are the same estimators in numpy)
# ---------------------------------------------------------------------------------------------
def _binary_curve(y, s):
    """Cumulative TP/FP at each distinct threshold, scores descending (sklearn
    `_binary_clf_curve` semantics: ties share one operating point)."""
    order = np.argsort(-s, kind='mergesort')
    y = y[order]
    s = s[order]
    last = np.r_[np.flatnonzero(np.diff(s)), y.size - 1]
    tps = np.cumsum(y, dtype=np.float64)[last]
    fps = 1.0 + last - tps
    return tps, fps


def auprc_auroc_ap(target_tensor, score_tensor):
    """(AUPRC, AUROC, AP) of one relation, as the reference computes them:
    AUROC = trapezoid over the ROC curve, AP = sum (R_n - R_{n-1}) P_n, AUPRC = trapezoid over the
    precision-recall curve (`metrics.auc(recall, precision)`)."""
    y = torch.as_tensor(target_tensor).detach().cpu().numpy().astype(np.float64).ravel()
    s = torch.as_tensor(score_tensor).detach().cpu().numpy().astype(np.float64).ravel()
    tps, fps = _binary_curve(y, s)
    P, N = tps[-1], fps[-1]
    if P == 0 or N == 0:
        raise ValueError('only one class present')
    tpr = np.r_[0.0, tps / P]
    fpr = np.r_[0.0, fps / N]
    auroc = float(np.sum(np.diff(fpr) * (tpr[1:] + tpr[:-1]) * 0.5))
    precision = tps / (tps + fps)
    recall = tps / P
    ap = float(np.sum(np.diff(np.r_[0.0, recall]) * precision))
    # precision_recall_curve: stop at full recall, reverse, append (recall 0, precision 1)
    stop = int(np.searchsorted(tps, P))
    pr = np.r_[precision[:stop + 1][::-1], 1.0]
    rc = np.r_[recall[:stop + 1][::-1], 0.0]
    auprc = float(-np.sum(np.diff(rc) * (pr[1:] + pr[:-1]) * 0.5))
    return auprc, auroc, ap


def auprc_auroc_ap_by_range(pos_score, neg_score, range_list):
    """[3, R] record of (AUPRC, AUROC, AP) per relation block (`src/layers.py:355-368`).
    Device tensors are evaluated by `tipk_rank_metrics` (one launch for all relations); host tensors,
    non-consecutive ranges or oversized relations take the numpy path below."""
    if torch.is_tensor(pos_score) and pos_score.is_cuda:
        rg = torch.as_tensor(range_list).to(torch.int64).cpu()
        consecutive = rg.numel() > 0 and int(rg[0, 0]) == 0 and bool((rg[1:, 0] == rg[:-1, 1]).all())
        if consecutive and int(rg[-1, 1]) == pos_score.numel() == neg_score.numel():
            from . import ops
            ptr = torch.cat([rg[:1, 0], rg[:, 1]]).to(pos_score.device)
            rec = ops.rank_metrics(pos_score.detach(), neg_score.detach(), ptr, int((rg[:, 1] - rg[:, 0]).max()))
            if rec is not None:
                return rec.cpu().numpy()
    pos = torch.as_tensor(pos_score).detach().cpu().numpy()
    neg = torch.as_tensor(neg_score).detach().cpu().numpy()
    rg = torch.as_tensor(range_list).cpu().numpy().astype(np.int64)
    rec = np.zeros((3, rg.shape[0]))
    for i, (a, b) in enumerate(rg):
        y = np.r_[np.ones(b - a), np.zeros(b - a)]
        s = np.r_[pos[a:b], neg[a:b]]
        rec[:, i] = auprc_auroc_ap(y, s)
    return rec
