#!/usr/bin/env python3
"""Re-encode the BioSNAP/Decagon graph shipped with the reference into one compact binary.

Runs ONLY in the build container (reads /root/reference/data, which does not exist on the GPU
box); the output `tip_amd/data/biosnap_v1.npz` (~10 MB) is committed so that bench/tests can load
the full graph anywhere.  It stores exactly what `prepare.py:5-16` + `data/utils.py:34-169,212-217`
feed into the random split, in the same element order, so `tip_amd.data.build_data_dict` can
replay `process_edges` / `process_prot_edge` (`src/utils.py:35-65`, `data/utils.py:212-229`) with
a seeded numpy generator and reproduce the `data_dict.pkl` schema:

  et_list   int32 [R]      relation-type ids (`data/decagon_et.pkl`, R = 1097)
  dd_ptr    int64 [R+1]    offsets into dd_pairs per relation
  dd_pairs  uint16 [2, P]  upper-triangular (row <= col) pairs of `sp.triu(adj).tocsr().tocoo()`
  pp_pairs  uint16 [2, Q]  `remove_bidirection([pp.col; pp.row])` i.e. pairs with [0] > [1]
  dp_pairs  uint16 [2, S]  `[dp.col - 1; dp.row - 1]` (protein, drug) in COO order of the CSR
  n_drug, n_prot           graph_info.pkl
"""
import pickle
import sys

import numpy as np
import scipy.sparse as sp

REF = sys.argv[1] if len(sys.argv) > 1 else '/root/reference/data/'
OUT = sys.argv[2] if len(sys.argv) > 2 else 'tip_amd/data/biosnap_v1.npz'


def main():
    with open(REF + 'decagon_et.pkl', 'rb') as f:
        et_list = pickle.load(f)
    with open(REF + 'graph_info.pkl', 'rb') as f:
        n_drug, n_prot, n_combo, n_mono = pickle.load(f)
    assert n_drug < 65536 and n_prot < 65536

    ptr = [0]
    rows, cols = [], []
    sum_adj = sp.csr_matrix((n_drug, n_drug))
    for t in et_list:
        adj = sp.load_npz('%ssym_adj/drug-sparse-adj/type_%d.npz' % (REF, t))
        sum_adj += adj
        coo = sp.triu(adj).tocsr().tocoo()                 # data/utils.py:60,151
        assert (coo.data == 1).all() and (coo.row < coo.col).all()
        rows.append(coo.row.astype(np.uint16))
        cols.append(coo.col.astype(np.uint16))
        ptr.append(ptr[-1] + coo.nnz)
    # data/utils.py:83-104 removes isolated drugs; with the shipped list there are none
    assert (np.asarray(sum_adj.sum(axis=1)).ravel() > 0).all(), "isolated drugs present"
    dd_pairs = np.stack([np.concatenate(rows), np.concatenate(cols)])

    pp = sp.load_npz(REF + 'sym_adj/protein-sparse-adj.npz').tocoo()       # data/utils.py:66,139
    idx = np.stack([pp.col, pp.row])                                       # data/utils.py:213-215
    keep = idx[0] > idx[1]                                                 # src/utils.py:8-9
    pp_pairs = idx[:, keep].astype(np.uint16)

    dp = sp.load_npz(REF + 'sym_adj/drug-protein-sparse-adj.npz').tocsr().tocoo()  # :71,138
    dp_pairs = np.stack([dp.col - 1, dp.row - 1])                          # prepare.py:30
    assert dp_pairs.min() >= 0
    dp_pairs = dp_pairs.astype(np.uint16)

    np.savez_compressed(OUT, et_list=np.asarray(et_list, np.int32), dd_ptr=np.asarray(ptr, np.int64),
                        dd_pairs=dd_pairs, pp_pairs=pp_pairs, dp_pairs=dp_pairs,
                        n_drug=np.int64(n_drug), n_prot=np.int64(n_prot))
    print('R=%d P=%d Q=%d S=%d -> %s' % (len(et_list), dd_pairs.shape[1], pp_pairs.shape[1],
                                        dp_pairs.shape[1], OUT))


if __name__ == '__main__':
    main()
