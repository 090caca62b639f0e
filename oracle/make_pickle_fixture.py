#!/usr/bin/env python3
"""Generate tests/golden/data_dict_small.pkl + tests/golden/tip_from_pickle.npz by running the REFERENCE's
own data pipeline and model.  TEST INFRASTRUCTURE; runs only in the build container.

north_star: "the reference's data_dict.pkl ... run unchanged".  The real file is 476 MB, so this script
feeds the reference's `prepare.py:10-44` -- executed FROM /root/reference/prepare.py at run time, not
restated, not stored -- a reduced copy of the reference's data directory:

  * 6 of the 1 097 relation files (`sym_adj/drug-sparse-adj/type_*.npz`, linked, not copied),
  * the protein graph and the drug-protein matrix restricted to the first 500 proteins,
  * graph_info.pkl with that protein count, the mono-feature matrix as is,

and pickles the resulting dict exactly as `prepare.py:46-47` does (scipy matrices, the per-relation
lists, `dd_y_pos/neg` and all).  Then the reference's `TIP(settings, device, data_path=...)`
(`src/layers.py:272-375`, imported unchanged over `oracle/pyg_restated`) is built on that file with
recorded weights; its training loss (with recorded negatives), all gradients and its `test()` record are
the golden values `tests/test_gpu_layers.py::test_reference_pickle_drop_in` checks the HIP `TIP` against
when it loads the SAME pickle through the same constructor.

    python oracle/make_pickle_fixture.py
"""
import os
import pickle
import shutil
import sys
import tempfile

import numpy as np
import scipy.sparse as sp
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = '/root/reference'
sys.path.insert(0, os.path.join(HERE, 'pyg_restated'))
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

N_PROT = 500
OUT_PKL = os.path.join(ROOT, 'tests', 'golden', 'data_dict_small.pkl')
OUT_NPZ = os.path.join(ROOT, 'tests', 'golden', 'tip_from_pickle.npz')


def reduced_data_dir(tmp, et_list):
    d = os.path.join(tmp, 'data')
    os.makedirs(os.path.join(d, 'sym_adj', 'drug-sparse-adj'))
    os.makedirs(os.path.join(d, 'node_feature'))
    drug_num, protein_num, combo_num, mono_num = pickle.load(open(os.path.join(REF, 'data', 'graph_info.pkl'), 'rb'))
    pickle.dump((drug_num, N_PROT, combo_num, mono_num), open(os.path.join(d, 'graph_info.pkl'), 'wb'))
    for i in et_list:
        os.symlink(os.path.join(REF, 'data', 'sym_adj', 'drug-sparse-adj', 'type_%d.npz' % i),
                   os.path.join(d, 'sym_adj', 'drug-sparse-adj', 'type_%d.npz' % i))
    os.symlink(os.path.join(REF, 'data', 'node_feature', 'drug-mono-feature.npz'),
               os.path.join(d, 'node_feature', 'drug-mono-feature.npz'))
    pp = sp.load_npz(os.path.join(REF, 'data', 'sym_adj', 'protein-sparse-adj.npz')).tocsr()
    sp.save_npz(os.path.join(d, 'sym_adj', 'protein-sparse-adj.npz'), pp[:N_PROT, :N_PROT].tocsr())
    dp = sp.load_npz(os.path.join(REF, 'data', 'sym_adj', 'drug-protein-sparse-adj.npz')).tocsr()
    # prepare.py:30 shifts both indices by -1, so protein column N_PROT is still a valid protein
    sp.save_npz(os.path.join(d, 'sym_adj', 'drug-protein-sparse-adj.npz'), dp[:, :N_PROT + 1].tocoo())
    return d


def run_reference_prepare(tmp, et_list):
    """exec the body of the reference's prepare.py (lines between the et_list load and the pickle dump)
    with cwd = tmp, so that its './data/' is the reduced directory."""
    src = open(os.path.join(REF, 'prepare.py')).read().splitlines()
    first = next(i for i, l in enumerate(src) if l.startswith('data = load_data_torch'))
    last = next(i for i, l in enumerate(src) if l.startswith('with open(out_file'))
    body = '\n'.join(src[first:last])
    ns = {}
    exec('from data.utils import load_data_torch, process_prot_edge\nfrom src.utils import *\nimport pickle\n', ns)
    ns['et_list'] = et_list
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        exec(compile(body, os.path.join(REF, 'prepare.py'), 'exec'), ns)
    finally:
        os.chdir(cwd)
    return ns['data']


def main():
    all_et = pickle.load(open(os.path.join(REF, 'data', 'decagon_et.pkl'), 'rb'))
    et_list = [all_et[i] for i in (0, 1, 2, 40, 400, 900)]
    tmp = tempfile.mkdtemp()
    try:
        reduced_data_dir(tmp, et_list)
        np.random.seed(1111)                                   # src/layers.py:14 (the split draws from it)
        torch.manual_seed(1111)
        data = run_reference_prepare(tmp, et_list)
    finally:
        shutil.rmtree(tmp)
    with open(OUT_PKL, 'wb') as f:                             # prepare.py:46-47
        pickle.dump(data, f)
    print('%s  %.1f KB  keys: %s' % (OUT_PKL, os.path.getsize(OUT_PKL) / 1024, sorted(data)))

    import src.layers as ref                                   # the reference, unchanged
    ref.device = torch.device('cpu')                           # src/layers.py:319 reads an undefined global
    np.random.seed(7)
    st = ref.Setting(sp_rate=0.9, lr=0.01, prot_drug_dim=16, n_embed=48, n_hid1=32, n_hid2=16, num_base=32)
    model = ref.TIP(st, torch.device('cpu'), data_path=OUT_PKL)        # tip.py:14-15
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():
        for prm in model.parameters():                         # zero biases etc. get values too
            prm.copy_(torch.randn(prm.shape, generator=g) * (0.3 if prm.dim() > 1 else 0.1))
    np.random.seed(8)
    neg = ref.typed_negative_sampling(model.data.dd_train_idx, model.data.n_drug, model.data.dd_train_range)
    np.random.seed(8)
    loss = model()
    loss.backward()
    rec = model.test(print_output=False)
    arrays = {'loss': loss.detach().numpy(), 'train_neg': neg.numpy(), 'test_neg': model.test_neg_index.numpy(),
              'record': np.asarray(rec), 'embeddings': model.embeddings.detach().numpy(),
              'n_train': np.asarray(model.data.dd_train_idx.shape[1]), 'n_drug': np.asarray(model.data.n_drug),
              'n_prot': np.asarray(model.data.n_prot), 'n_dd_et': np.asarray(model.data.n_dd_et)}
    for k, v in model.state_dict().items():
        arrays['param.' + k] = v.detach().numpy()
    for k, v in model.named_parameters():
        arrays['grad.' + k] = v.grad.detach().numpy()
    np.savez_compressed(OUT_NPZ, **arrays)
    print('%s  %.1f KB  loss %.6f  n_drug %d n_prot %d R %d train edges %d' % (
        OUT_NPZ, os.path.getsize(OUT_NPZ) / 1024, float(loss), model.data.n_drug, model.data.n_prot,
        model.data.n_dd_et, model.data.dd_train_idx.shape[1]))


if __name__ == '__main__':
    main()
