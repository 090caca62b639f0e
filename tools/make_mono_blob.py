#!/usr/bin/env python3
"""Re-encode the reference's mono side-effect drug features (`data/node_feature/drug-mono-feature.npz`,
645 x 10 184, 174 977 ones -- the "TODO: add drug feature" of prepare.py:21) into the compact
`tip_amd/data/biosnap_mono_v1.npz`.  Runs ONLY in the build container.

  mono_pairs  uint16 [2, M]  (drug, feature column) in the COO order of `drug_mono_adj.tocsr().tocoo()`,
                             which is the order `data/utils.py:117-132` appends them to the identity block
  n_drug, n_mono
"""
import sys

import numpy as np
import scipy.sparse as sp

REF = sys.argv[1] if len(sys.argv) > 1 else '/root/reference/data/'
OUT = sys.argv[2] if len(sys.argv) > 2 else 'tip_amd/data/biosnap_mono_v1.npz'

m = sp.load_npz(REF + 'node_feature/drug-mono-feature.npz').tocsr().tocoo()      # data/utils.py:76-78,124
assert (m.data == 1).all() and m.shape[1] < 65536
np.savez_compressed(OUT, mono_pairs=np.stack([m.row, m.col]).astype(np.uint16), n_drug=np.int64(m.shape[0]),
                    n_mono=np.int64(m.shape[1]))
print('%d x %d, %d entries -> %s' % (m.shape[0], m.shape[1], m.nnz, OUT))
