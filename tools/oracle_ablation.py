#!/usr/bin/env python3
"""TEST TOOL (not product, CPU only): which documented difference carries the gap between this path's test AUPRC and
the only accuracy figure the reference publishes (analysis/evaluation.ipynb:192-195: TIP AUPRC 0.948, R = 963, 100 epochs)?

The published figure was not produced by `tip.py` / `src/layers.py` (the path this repository rebuilds) but by the
`model/*.py` family of scripts (`model/ddm-df_rgcn.py:35-61`): those encoders end with a ReLU on the embeddings
(`F.relu(x, inplace=True)` after rgcn2, :59), which `FMEncoder.forward` (`src/layers.py:545-549`) does not have.
This tool trains the CPU ORACLE alone (tests/parity_harness.py `oracle_step`, i.e. the arithmetic the HIP path is
held to) and toggles one difference at a time:

    --final-relu       z = relu(rgcn2(...)) as in model/ddm-df_rgcn.py:59
    --ref-sampler      the reference's own sampler incl. its leaking resample loop (src/neg_sampling.py:5-19, numpy RNG)
                       instead of the Philox spec of the device sampler (exact rejection)
    --n-embed 16 --num-base 16   the widths of the published run's name ('fm-(32-16)-(16-16-32-32-16)', evaluation.ipynb cell 11)
    --min-pairs 500    the paper's 963 relations
    --epochs 100

Writes one JSON line (and profiles/<tag>.json with --tag)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import tip_oracle as O                                       # noqa: E402
from oracle import philox_sampler as PS                                  # noqa: E402
from parity_harness import OracleAdam                                    # noqa: E402
from tip_amd.data import build_data_dict                                 # noqa: E402
from tip_amd.utils import auprc_auroc_ap_by_range                        # noqa: E402


def step(po, dd, mod, neg, final_relu):
    enc_p = {k: v for k, v in po.items() if k != 'decoder.weight'}
    z0, saved = O.fm_encoder_fwd(enc_p, dd, mod)
    z = torch.relu(z0) if final_relu else z0
    w = po['decoder.weight']
    ps = O.distmult_fwd(z, dd['dd_train_idx'], dd['dd_train_et'], w)
    ns = O.distmult_fwd(z, neg, dd['dd_train_et'], w)
    lo = O.tip_loss(ps, ns)
    gp, gn = O.tip_loss_bwd(ps, ns)
    gz1, gw1 = O.distmult_bwd(gp, z, dd['dd_train_idx'], dd['dd_train_et'], w)
    gz2, gw2 = O.distmult_bwd(gn, z, neg, dd['dd_train_et'], w)
    gz = gz1 + gz2
    if final_relu:
        gz = gz * (z0 > 0).to(gz.dtype)
    grads = O.fm_encoder_bwd(gz, enc_p, dd, saved, mod)
    grads['decoder.weight'] = gw1 + gw2
    return lo, grads, z


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--epochs', type=int, default=100)
    ap.add_argument('--min-pairs', type=int, default=None)
    ap.add_argument('--relations', type=int, default=None)
    ap.add_argument('--mod', default='cat')
    ap.add_argument('--final-relu', action='store_true')
    ap.add_argument('--ref-sampler', action='store_true')
    ap.add_argument('--n-embed', type=int, default=None)
    ap.add_argument('--num-base', type=int, default=32)
    ap.add_argument('--threads', type=int, default=4)
    ap.add_argument('--tag', default=None)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    dd = build_data_dict(min_pairs=a.min_pairs, max_relations=a.relations)
    R = dd['n_dd_et']
    dims = dict(prot_drug_dim=16, n_embed=48) if a.mod == 'cat' else dict(prot_drug_dim=64, n_embed=64)
    if a.n_embed is not None:
        dims['n_embed'] = a.n_embed
    po = O.init_params(dd['n_drug'], dd['n_prot'], R, mod=a.mod, seed=1111, num_base=a.num_base, **dims)
    opt = OracleAdam(po, 0.01)
    rng = np.random.RandomState(1111)
    rel_ptr = np.concatenate([[0], np.asarray(dd['dd_train_range'])[:, 1]]).astype(np.int64)
    pos_np = dd['dd_train_idx'].numpy()
    te_ptr = np.concatenate([[0], np.asarray(dd['dd_test_range'])[:, 1]]).astype(np.int64)
    n = dd['n_drug']

    def sample(pos_t, pos_n, ptr, rg, call):
        if a.ref_sampler:
            return O.typed_negative_sampling(pos_t, n, rg, rng)
        return torch.from_numpy(PS.typed_negative_sampling_spec(pos_n, n, ptr, PS.call_key(1111, call)))
    test_neg = sample(dd['dd_test_idx'], dd['dd_test_idx'].numpy(), te_ptr, dd['dd_test_range'], 0)
    t0 = time.time()
    losses = []
    z = None
    for ep in range(a.epochs):
        neg = sample(dd['dd_train_idx'], pos_np, rel_ptr, dd['dd_train_range'], ep + 1)
        lo, grads, z = step(po, dd, a.mod, neg, a.final_relu)
        opt.step(grads)
        losses.append(float(lo))
        if ep % 10 == 0 or ep == a.epochs - 1:
            print('epoch %3d loss %.6f  (%.0f s)' % (ep, losses[-1], time.time() - t0), flush=True)
    w = po['decoder.weight']
    ps = O.distmult_fwd(z, dd['dd_test_idx'], dd['dd_test_et'], w)
    ns = O.distmult_fwd(z, test_neg, dd['dd_test_et'], w)
    rec = auprc_auroc_ap_by_range(ps, ns, dd['dd_test_range'])
    out = {'tool': 'oracle_ablation', 'epochs': a.epochs, 'relations': R, 'mod': a.mod, 'final_relu': a.final_relu,
           'ref_sampler': a.ref_sampler, 'n_embed': dims['n_embed'], 'num_base': a.num_base, 'loss_first': losses[0], 'loss_last': losses[-1],
           'oracle': dict(zip(['auprc', 'auroc', 'ap'], (rec.sum(1) / R).tolist())),
           'reference_published_auprc': 0.948, 'reference_published_where': 'analysis/evaluation.ipynb:192-195 (model/*.py scripts, R = 963)',
           'train_edges': int(dd['dd_train_idx'].shape[1]), 's_total': time.time() - t0, 'threads': a.threads}
    print(json.dumps(out))
    if a.tag:
        json.dump(out, open(os.path.join(ROOT, 'profiles', a.tag + '.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
