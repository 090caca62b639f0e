"""TEST INFRASTRUCTURE: lock-step training of the HIP path and the CPU oracle (same split, same initial
weights, same negatives every epoch, Adam(lr=0.01), full batch -- reference tip.py:14-30,
src/layers.py:328-342).  Used by tests/test_gpu_train_parity.py and tools/auroc_parity.py; the oracle is
the checker here, never part of the product path."""
import os
import time

import numpy as np
import torch

from oracle import tip_oracle as O


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


def oracle_step(po, dd, mod, neg_cpu):
    """One oracle training step on parameter dict `po` -> (loss, grads, z)."""
    enc_p = {k: v for k, v in po.items() if k != 'decoder.weight'}
    z, saved = O.fm_encoder_fwd(enc_p, dd, mod)
    w = po['decoder.weight']
    ps = O.distmult_fwd(z, dd['dd_train_idx'], dd['dd_train_et'], w)
    ns = O.distmult_fwd(z, neg_cpu, dd['dd_train_et'], w)
    lo = O.tip_loss(ps, ns)
    gp, gn = O.tip_loss_bwd(ps, ns)
    gz1, gw1 = O.distmult_bwd(gp, z, dd['dd_train_idx'], dd['dd_train_et'], w)
    gz2, gw2 = O.distmult_bwd(gn, z, neg_cpu, dd['dd_train_et'], w)
    grads = O.fm_encoder_bwd(gz1 + gz2, enc_p, dd, saved, mod)
    grads['decoder.weight'] = gw1 + gw2
    return lo, grads, z


def run_parity(dd, mod='cat', epochs=10, dev='cuda:0', threads=16, snapshots=(), log=None):
    """Train `epochs` full-batch epochs on both sides.  snapshots: epochs (1-based) after whose optimizer
    step the parameters and embeddings of both sides are recorded.  -> dict."""
    from tip_amd.layers import Setting, TIP
    from tip_amd.neg_sampling import typed_negative_sampling
    from tip_amd.utils import auprc_auroc_ap_by_range
    torch.set_num_threads(max(1, min(threads, os.cpu_count() or 1)))
    dev = torch.device(dev)
    R = dd['n_dd_et']
    dims = dict(prot_drug_dim=16, n_embed=48) if mod == 'cat' else dict(prot_drug_dim=64, n_embed=64)
    st = Setting(sp_rate=0.9, lr=0.01, n_hid1=32, n_hid2=16, num_base=32, **dims)
    p = O.init_params(dd['n_drug'], dd['n_prot'], R, mod=mod, seed=1111, **dims)
    model = TIP(st, dev, mod=mod, data=dd)
    sd = model.state_dict()
    for k in sd:
        sd[k] = p[k[len('encoder.'):]].clone() if k.startswith('encoder.') else p[k].clone()
    model.load_state_dict(sd)
    from tip_amd.optim import Adam                     # the product's optimizer step (tipk_adam_step), checked against OracleAdam
    opt = Adam(model.parameters(), lr=st.lr)
    po = {k: v.clone() for k, v in p.items()}
    oopt = OracleAdam(po, st.lr)
    d = model.data
    test_neg = model.test_neg_index.cpu()
    t_gpu = t_cpu = 0.0
    hist, snaps = [], {}
    z = None
    for ep in range(epochs):
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
        lo, grads, z = oracle_step(po, dd, mod, neg.cpu())
        oopt.step(grads)
        t_cpu += time.perf_counter() - t0
        hist.append((float(loss.detach()), float(lo)))
        if (ep + 1) in snapshots:
            hip = {k[len('encoder.'):] if k.startswith('encoder.') else k: v.detach().cpu().clone()
                   for k, v in model.state_dict().items()}
            snaps[ep + 1] = {'hip': hip, 'oracle': {k: v.clone() for k, v in po.items()},
                             'z_hip': model.embeddings.detach().cpu().clone(), 'z_oracle': z.clone()}
        if log and (ep % 10 == 0 or ep == epochs - 1):
            log('epoch %3d  loss hip %.6f  oracle %.6f' % (ep, hist[-1][0], hist[-1][1]))
    # evaluation exactly as TIP.test(): embeddings of the last training forward, fixed test negatives
    rec_gpu = model.test(print_output=False)
    w = po['decoder.weight']
    ps = O.distmult_fwd(z, dd['dd_test_idx'], dd['dd_test_et'], w)
    ns = O.distmult_fwd(z, test_neg, dd['dd_test_et'], w)
    rec_cpu = auprc_auroc_ap_by_range(ps, ns, dd['dd_test_range'])
    return {'epochs': epochs, 'relations': R, 'mod': mod, 'train_edges': int(dd['dd_train_idx'].shape[1]),
            'loss': hist, 'snapshots': snaps, 'rec_hip': rec_gpu, 'rec_oracle': rec_cpu,
            'hip': dict(zip(['auprc', 'auroc', 'ap'], (rec_gpu.sum(1) / R).tolist())),
            'oracle': dict(zip(['auprc', 'auroc', 'ap'], (rec_cpu.sum(1) / R).tolist())),
            'abs_diff_auroc': abs(float(rec_gpu[1].mean() - rec_cpu[1].mean())),
            'max_rel_auroc_diff_per_relation': float(np.abs(rec_gpu[1] - rec_cpu[1]).max()),
            'ms_per_epoch_hip_full_step': t_gpu / epochs * 1e3, 's_per_epoch_oracle_cpu': t_cpu / epochs,
            'cpu_threads': torch.get_num_threads()}
