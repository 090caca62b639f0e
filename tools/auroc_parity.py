#!/usr/bin/env python3
"""End-to-end training parity: the HIP path vs the CPU oracle, same split, same initial weights,
same negatives every epoch, Adam(lr=0.01), full batch (reference tip.py:14-30).

north_star criterion: AUROC within +-0.002 of the reference path.  The literal reference needs
~470 s/epoch on CPU, so -- as SURVEY.md section 8(c) prescribes -- the 100-epoch
reference value comes from the oracle (validated against the reference's own code by the golden
fixtures).  The negatives are drawn by the device sampler and handed to both sides.

    python tools/auroc_parity.py --epochs 100 [--relations 1097] [--mod cat]
Writes one JSON line (also to profiles/<tag>_auroc_parity.json with --tag).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import tip_oracle as O                                      # noqa: E402
from tip_amd.data import build_data_dict                                # noqa: E402
from tip_amd.layers import Setting, TIP                                 # noqa: E402
from tip_amd.neg_sampling import typed_negative_sampling                # noqa: E402
from tip_amd.utils import auprc_auroc_ap_by_range                       # noqa: E402


class OracleAdam(object):
    """torch.optim.Adam defaults (betas .9/.999, eps 1e-8, no weight decay) on a dict of tensors."""

    def __init__(self, params, lr):
        self.p, self.lr, self.t = params, lr, 0
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}

    def step(self, grads):
        self.t += 1
        b1, b2 = 0.9, 0.999
        for k, g in grads.items():
            self.m[k].mul_(b1).add_(g, alpha=1 - b1)
            self.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
            mhat = self.m[k] / (1 - b1 ** self.t)
            vhat = self.v[k] / (1 - b2 ** self.t)
            self.p[k].sub_(self.lr * mhat / (vhat.sqrt() + 1e-8))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--epochs', type=int, default=100)
    ap.add_argument('--relations', type=int, default=None)
    ap.add_argument('--mod', default='cat')
    ap.add_argument('--threads', type=int, default=16)
    ap.add_argument('--tag', default=None)
    args = ap.parse_args()
    torch.set_num_threads(min(args.threads, os.cpu_count() or 1))
    dev = torch.device('cuda:0')
    dd = build_data_dict(max_relations=args.relations)
    R = dd['n_dd_et']
    dims = dict(prot_drug_dim=16, n_embed=48) if args.mod == 'cat' else dict(prot_drug_dim=64, n_embed=64)
    st = Setting(sp_rate=0.9, lr=0.01, n_hid1=32, n_hid2=16, num_base=32, **dims)
    p = O.init_params(dd['n_drug'], dd['n_prot'], R, mod=args.mod, seed=1111, **dims)

    model = TIP(st, dev, mod=args.mod, data=dd)
    sd = model.state_dict()
    for k in sd:
        sd[k] = p[k[len('encoder.'):]].clone() if k.startswith('encoder.') else p[k].clone()
    model.load_state_dict(sd)
    opt = torch.optim.Adam(model.parameters(), lr=st.lr)
    po = {k: v.clone() for k, v in p.items()}
    oopt = OracleAdam(po, st.lr)
    d = model.data
    test_neg = model.test_neg_index.cpu()

    t_gpu = t_cpu = 0.0
    hist = []
    for ep in range(args.epochs):
        neg = typed_negative_sampling(d.dd_train_idx, d.n_drug, d.dd_train_range)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        opt.zero_grad()
        loss = model(neg_index=neg)
        loss.backward()
        opt.step()
        torch.cuda.synchronize()
        t_gpu += time.perf_counter() - t0

        t0 = time.perf_counter()
        negc = neg.cpu()
        enc_p = {k: v for k, v in po.items() if k != 'decoder.weight'}
        z, saved = O.fm_encoder_fwd(enc_p, dd, args.mod)
        w = po['decoder.weight']
        ps = O.distmult_fwd(z, dd['dd_train_idx'], dd['dd_train_et'], w)
        ns = O.distmult_fwd(z, negc, dd['dd_train_et'], w)
        lo = O.tip_loss(ps, ns)
        gp, gn = O.tip_loss_bwd(ps, ns)
        gz1, gw1 = O.distmult_bwd(gp, z, dd['dd_train_idx'], dd['dd_train_et'], w)
        gz2, gw2 = O.distmult_bwd(gn, z, negc, dd['dd_train_et'], w)
        grads = O.fm_encoder_bwd(gz1 + gz2, enc_p, dd, saved, args.mod)
        grads['decoder.weight'] = gw1 + gw2
        oopt.step(grads)
        t_cpu += time.perf_counter() - t0
        hist.append((float(loss), float(lo)))
        if ep % 10 == 0 or ep == args.epochs - 1:
            print('epoch %3d  loss hip %.6f  oracle %.6f' % (ep, hist[-1][0], hist[-1][1]), flush=True)

    # evaluation exactly as TIP.test(): embeddings of the last training forward, fixed test negatives
    rec_gpu = model.test(print_output=False)
    ps = O.distmult_fwd(z, dd['dd_test_idx'], dd['dd_test_et'], w)
    ns = O.distmult_fwd(z, test_neg, dd['dd_test_et'], w)
    rec_cpu = auprc_auroc_ap_by_range(ps, ns, dd['dd_test_range'])
    out = {'epochs': args.epochs, 'relations': R, 'mod': args.mod, 'train_edges': int(dd['dd_train_idx'].shape[1]),
           'loss_first': hist[0], 'loss_last': hist[-1],
           'hip': dict(zip(['auprc', 'auroc', 'ap'], (rec_gpu.sum(1) / R).tolist())),
           'oracle': dict(zip(['auprc', 'auroc', 'ap'], (rec_cpu.sum(1) / R).tolist())),
           'abs_diff_auroc': abs(float(rec_gpu[1].mean() - rec_cpu[1].mean())),
           'max_rel_auroc_diff_per_relation': float(np.abs(rec_gpu[1] - rec_cpu[1]).max()),
           'ms_per_epoch_hip_full_step': t_gpu / args.epochs * 1e3,
           's_per_epoch_oracle_cpu': t_cpu / args.epochs, 'cpu_threads': torch.get_num_threads()}
    print(json.dumps(out))
    if args.tag:
        os.makedirs('profiles', exist_ok=True)
        json.dump(out, open('profiles/%s_auroc_parity.json' % args.tag, 'w'), indent=1)


if __name__ == '__main__':
    main()
