#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own code.  TEST INFRASTRUCTURE.

Runs only in the build container: it puts /root/reference and `oracle/pyg_restated` (the restated
PyG 2.0.1 subset, see its docstring) on sys.path, imports `src/layers.py` UNCHANGED, feeds it small
seeded graphs plus explicit weights, and records inputs, outputs and every parameter gradient
(autograd of the reference).  The fixtures are data only; no reference source is stored.

    python oracle/make_golden.py            # rewrites tests/golden/
    python oracle/make_golden.py drug_features check_loader     # only the named fixture(s)
    python oracle/make_golden.py fast_route                     # round 6: the pair-form shapes (tip.py:14 dims)
"""
import os
import pickle
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, 'pyg_restated'))
sys.path.insert(0, '/root/reference')
sys.path.insert(0, ROOT)

import src.layers as ref                      # noqa: E402  (the reference, unchanged)
from src.utils import sparse_id as ref_sparse_id, process_edges as ref_process_edges  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')


def npz(name, **arrays):
    conv = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **conv)
    print('%-28s %7.1f KB' % (name, os.path.getsize(path) / 1024))


# ---------------------------------------------------------------------------------------------
# small seeded graphs with the nasty cases of SURVEY.md section 8(c)
# ---------------------------------------------------------------------------------------------
def small_graph(seed, n_drug=37, n_prot=211, sizes=(40, 1, 6, 90, 3, 17, 250), symmetric=True, nasty=False):
    rng = np.random.RandomState(seed)
    blocks = []
    for s in sizes:
        # drugs 0..n_drug-4 only: the last three drugs have zero in-degree; sampling with
        # replacement gives duplicate edges in the big relations
        u = rng.randint(0, n_drug - 3, s)
        v = rng.randint(0, n_drug - 3, s)
        if nasty and s >= 9:
            # explicit self pairs (u == v: on a symmetric graph the mirrored half repeats them) and an explicit duplicate edge
            # inside the relation, whatever the draw gave
            v[0] = u[0]
            v[3] = u[3]
            u[2], v[2] = u[1], v[1]
        e = torch.from_numpy(np.stack([u, v]).astype(np.int64))
        blocks.append(torch.cat([e, e.flip(0)], dim=1) if symmetric else e)
    dd_idx = torch.cat(blocks, dim=1)
    nb = [b.shape[1] for b in blocks]
    dd_et = torch.repeat_interleave(torch.arange(len(blocks)), torch.tensor(nb))
    end = torch.cumsum(torch.tensor(nb), 0)
    dd_range = torch.stack([end - torch.tensor(nb), end], dim=1)
    # P-P: random pairs, mirrored, plus explicit self loops and a duplicate edge
    a = rng.randint(0, n_prot - 5, (2, 600))            # last proteins isolated
    pp = np.concatenate([a, a[::-1], np.array([[3, 9, 9], [3, 9, 9]]), a[:, :4]], axis=1)
    pp_idx = torch.from_numpy(pp.astype(np.int64))
    # P->D: (protein, n_prot + drug), sorted by drug; about a third of the drugs have targets
    dr = np.sort(rng.choice(np.arange(0, n_drug, 3), 60))
    pr = rng.randint(0, n_prot, 60)
    dp_idx = torch.from_numpy(np.stack([pr, dr + n_prot]).astype(np.int64))
    d_norm = torch.from_numpy(rng.uniform(0.5, 2.0, n_drug).astype(np.float32))
    return dict(dd_idx=dd_idx, dd_et=dd_et, dd_range=dd_range, pp_idx=pp_idx, dp_idx=dp_idx,
                d_norm=d_norm, n_drug=n_drug, n_prot=n_prot, n_rel=len(sizes))


def randomize_(module, seed):
    """Explicit weights (also non-zero biases) so fixtures do not depend on init RNG streams."""
    g = torch.Generator().manual_seed(seed)
    for _, prm in sorted(module.named_parameters()):
        prm.data = torch.randn(prm.shape, generator=g) * (0.5 if prm.dim() > 1 else 0.1)


def params_of(module):
    return {k: v.detach().clone() for k, v in module.named_parameters()}


def grads_of(module):
    return {'grad.' + k: v.grad.detach().clone() for k, v in module.named_parameters()}


# ---------------------------------------------------------------------------------------------
def golden_rgcn(seed, d_in, d_out, n_base, symmetric, name):
    g = small_graph(seed, symmetric=symmetric)
    torch.manual_seed(seed)
    x = torch.randn(g['n_drug'], d_in, requires_grad=True)
    up = torch.randn(g['n_drug'], d_out)
    for cls, tag in ((ref.MyRGCNConv2, 'conv2'), (ref.MyRGCNConv, 'conv')):
        m = cls(d_in, d_out, g['n_rel'], n_base, after_relu=False)
        randomize_(m, seed + 1)
        x.grad = None
        if tag == 'conv2':
            out = m(x, g['dd_idx'], g['dd_et'], g['dd_range'])
        else:
            out = m(x, g['dd_idx'], g['dd_et'])
        (out * up).sum().backward()
        if tag == 'conv2':
            keep = dict(out=out, grad_x=x.grad.clone(), **params_of(m), **grads_of(m))
        else:   # the bmm variant must agree with the range variant
            assert torch.allclose(out, keep['out'], atol=1e-4), 'MyRGCNConv != MyRGCNConv2'
            keep['out_bmm_variant'] = out
    npz(name, x=x, upstream=up, dd_idx=g['dd_idx'], dd_et=g['dd_et'], dd_range=g['dd_range'], **keep)


def golden_rgcn_fast(seed, symmetric, name):
    """The two R-GCN layers at the dims of tip.py:14 (64 -> 32 -> 16, num_base 32) on <= 64 drugs with self pairs, duplicate
    edges inside a relation, one- and two-edge relations and isolated drugs: the shapes the PAIR-FORM route of the build takes
    (tip_amd/layers.py `rgcn_graph`: n_bases = 32, d in {32, 16}), which the small-dims fixtures above never reach."""
    g = small_graph(seed, n_drug=61, sizes=(300, 1, 40, 700, 9, 120, 1500, 2, 64), symmetric=symmetric, nasty=True)
    assert bool((g['dd_idx'][0] == g['dd_idx'][1]).any())
    torch.manual_seed(seed)
    x = torch.randn(g['n_drug'], 64, requires_grad=True)
    up = torch.randn(g['n_drug'], 16)
    m1 = ref.MyRGCNConv2(64, 32, g['n_rel'], 32, after_relu=False)
    m2 = ref.MyRGCNConv2(32, 16, g['n_rel'], 32, after_relu=True)
    randomize_(m1, seed + 1)
    randomize_(m2, seed + 2)
    h = m1(x, g['dd_idx'], g['dd_et'], g['dd_range'])
    h.retain_grad()
    out = m2(torch.relu(h), g['dd_idx'], g['dd_et'], g['dd_range'])
    (out * up).sum().backward()
    npz(name, x=x, upstream=up, dd_idx=g['dd_idx'], dd_et=g['dd_et'], dd_range=g['dd_range'], hidden=h, out=out, grad_x=x.grad,
        grad_hidden=h.grad, **{'l1.' + k: v for k, v in params_of(m1).items()}, **{'l2.' + k: v for k, v in params_of(m2).items()},
        **{'grad.l1.' + k[5:]: v for k, v in grads_of(m1).items()}, **{'grad.l2.' + k[5:]: v for k, v in grads_of(m2).items()})


def golden_encoder_fast(seed, mod, symmetric, name):
    """FMEncoder at the dims of tip.py:14 / :17 on the graph of `golden_rgcn_fast`: the whole fast route of the build (layer
    hand-over, both layers' pair cells in one launch, pair-form backward) against the reference's own forward + autograd."""
    g = small_graph(seed, n_drug=61, sizes=(300, 1, 40, 700, 9, 120, 1500, 2, 64), symmetric=symmetric, nasty=True)
    torch.manual_seed(seed)
    kw = dict(prot_drug_dim=16, num_base=32, n_embed=48, n_hid1=32, n_hid2=16) if mod == 'cat' else \
        dict(prot_drug_dim=64, num_base=32, n_embed=64, n_hid1=32, n_hid2=16)
    enc = ref.FMEncoder('cpu', g['n_drug'], g['n_rel'], g['n_prot'], g['n_prot'], g['n_drug'], mod=mod, **kw)
    randomize_(enc, seed)
    up = torch.randn(g['n_drug'], 16)
    z = enc(ref_sparse_id(g['n_drug']), g['dd_idx'], g['dd_et'], g['dd_range'], g['d_norm'],
            ref_sparse_id(g['n_prot']), g['pp_idx'], g['dp_idx'], None)
    (z * up).sum().backward()
    npz(name, z=z, upstream=up, mod=mod, **{k: v for k, v in g.items()}, **params_of(enc),
        **grads_of(enc), **{'cfg.' + k: v for k, v in kw.items()})


def golden_hier(seed):
    g = small_graph(seed)
    torch.manual_seed(seed)
    m = ref.MyHierarchyConv(16, 8, g['n_prot'], g['n_drug'])
    randomize_(m, seed)
    x = torch.randn(g['n_prot'] + g['n_drug'], 16, requires_grad=True)
    up = torch.randn(g['n_drug'], 8)
    out = m(x, g['dp_idx'], None)
    (out * up).sum().backward()
    npz('hier_conv', x=x, upstream=up, dp_idx=g['dp_idx'], n_source=g['n_prot'], out=out,
        grad_x=x.grad, **params_of(m), **grads_of(m))


def golden_pp(seed):
    g = small_graph(seed)
    torch.manual_seed(seed)
    m = ref.PPEncoder(g['n_prot'])
    randomize_(m, seed)
    up = torch.randn(g['n_prot'], 16)
    out = m(ref_sparse_id(g['n_prot']), g['pp_idx'])
    (out * up).sum().backward()
    npz('pp_encoder', pp_idx=g['pp_idx'], n_prot=g['n_prot'], upstream=up, out=out,
        **params_of(m), **grads_of(m))
    # dense (non-identity) features through the same layers
    m2 = ref.PPEncoder(24)
    randomize_(m2, seed + 5)
    xd = torch.randn(g['n_prot'], 24, requires_grad=True)
    out2 = m2(xd, g['pp_idx'])
    (out2 * up).sum().backward()
    npz('pp_encoder_dense', pp_idx=g['pp_idx'], x=xd, upstream=up, out=out2, grad_x=xd.grad,
        **params_of(m2), **grads_of(m2))


def golden_decoder(seed):
    g = small_graph(seed)
    torch.manual_seed(seed)
    m = ref.MultiInnerProductDecoder(4, g['n_rel'])
    randomize_(m, seed)
    z = torch.randn(g['n_drug'], 4, requires_grad=True)
    rec = {}
    for sig in (True, False):
        z.grad = None
        m.weight.grad = None
        s = m(z, g['dd_idx'], g['dd_et'], sigmoid=sig)
        up = torch.linspace(-1, 1, s.numel())
        (s * up).sum().backward()
        rec['score_%d' % sig] = s
        rec['grad_z_%d' % sig] = z.grad.clone()
        rec['grad_weight_%d' % sig] = m.weight.grad.clone()
        rec['upstream'] = up
    # loss of src/layers.py:338-340 on (pos, pseudo-neg) scores
    z.grad = None
    m.weight.grad = None
    neg_idx = torch.stack([g['dd_idx'][0].flip(0), (g['dd_idx'][1] * 7 + 3) % g['n_drug']])
    pos = m(z, g['dd_idx'], g['dd_et'])
    neg = m(z, neg_idx, g['dd_et'])
    loss = -torch.log(pos + ref.EPS).mean() - torch.log(1 - neg + ref.EPS).mean()
    loss.backward()
    npz('decoder', z=z, weight=m.weight, dd_idx=g['dd_idx'], dd_et=g['dd_et'], neg_idx=neg_idx,
        loss=loss, loss_grad_z=z.grad, loss_grad_weight=m.weight.grad, **rec)


def golden_nn_decoder(seed):
    g = small_graph(seed)
    torch.manual_seed(seed)
    m = ref.NNDecoder(6, g['n_rel'], l1_dim=5)
    randomize_(m, seed)
    z = torch.randn(g['n_drug'], 6, requires_grad=True)
    s = m(z, g['dd_idx'], g['dd_et'])
    up = torch.linspace(-1, 1, s.numel())
    (s * up).sum().backward()
    npz('nn_decoder', z=z, dd_idx=g['dd_idx'], dd_et=g['dd_et'], upstream=up, score=s, grad_z=z.grad,
        **params_of(m), **grads_of(m))


def golden_encoder(seed, mod, name):
    g = small_graph(seed)
    torch.manual_seed(seed)
    kw = dict(prot_drug_dim=8, num_base=5, n_embed=12 if mod == 'cat' else 8, n_hid1=8, n_hid2=4)
    enc = ref.FMEncoder('cpu', g['n_drug'], g['n_rel'], g['n_prot'], g['n_prot'], g['n_drug'],
                        mod=mod, **kw)
    randomize_(enc, seed)
    up = torch.randn(g['n_drug'], 4)
    z = enc(ref_sparse_id(g['n_drug']), g['dd_idx'], g['dd_et'], g['dd_range'], g['d_norm'],
            ref_sparse_id(g['n_prot']), g['pp_idx'], g['dp_idx'], None)
    (z * up).sum().backward()
    npz(name, z=z, upstream=up, mod=mod, **{k: v for k, v in g.items()}, **params_of(enc),
        **grads_of(enc), **{'cfg.' + k: v for k, v in kw.items()})


def golden_tip(seed):
    """TIP end to end on a small data_dict: construction (incl. the initial encoder pass and the
    fixed test negatives), one `forward()` -> loss, backward, and `test()`."""
    rng = np.random.RandomState(seed)
    n_drug, n_prot = 41, 97
    raw = []
    for s in (30, 25, 120, 60):
        u = rng.randint(0, n_drug, s)
        v = rng.randint(0, n_drug, s)
        raw.append(torch.from_numpy(np.stack([np.minimum(u, v), np.maximum(u, v) + 0]).astype(np.int64)))
    np.random.seed(seed)
    d = {}
    (d['dd_train_idx'], d['dd_train_et'], d['dd_train_range'], d['dd_test_idx'], d['dd_test_et'],
     d['dd_test_range']) = ref_process_edges(raw)
    a = rng.randint(0, n_prot, (2, 300))
    d['pp_train_indices'] = torch.from_numpy(np.concatenate([a, a[::-1]], 1).astype(np.int64))
    dr = np.sort(rng.randint(0, n_drug, 50))
    pr = rng.randint(0, n_prot, 50)
    d['dp_edge_index'] = torch.from_numpy(np.stack([pr, dr + n_prot]).astype(np.int64))
    d['dp_range_list'] = torch.zeros((n_drug, 2))
    d['d_feat'], d['p_feat'] = ref_sparse_id(n_drug), ref_sparse_id(n_prot)
    d['n_drug'], d['n_prot'], d['n_dd_et'], d['n_drug_feat'] = n_drug, n_prot, len(raw), n_drug
    d['d_norm'] = torch.ones(n_drug)
    with tempfile.NamedTemporaryFile(suffix='.pkl', delete=False) as f:
        pickle.dump(d, f)
        path = f.name
    ref.device = torch.device('cpu')          # src/layers.py:319 reads an undefined global
    np.random.seed(seed + 1)
    st = ref.Setting(sp_rate=0.9, lr=0.01, prot_drug_dim=8, n_embed=8, n_hid1=8, n_hid2=4, num_base=3)
    model = ref.TIP(st, torch.device('cpu'), mod='add', data_path=path)
    os.unlink(path)
    randomize_(model, seed)
    # replay forward() (:328-342) with the negatives it would draw, so they can be recorded
    np.random.seed(seed + 2)
    neg = ref.typed_negative_sampling(model.data.dd_train_idx, n_drug, model.data.dd_train_range)
    np.random.seed(seed + 2)
    loss = model()
    loss.backward()
    rec = model.test(print_output=False)
    keep = {k: v for k, v in d.items() if isinstance(v, torch.Tensor) and not v.is_sparse}
    npz('tip_add_small', loss=loss, train_neg=neg, test_neg=model.test_neg_index, record=rec,
        embeddings=model.embeddings, n_drug=n_drug, n_prot=n_prot, n_dd_et=len(raw),
        **keep, **params_of(model), **grads_of(model))


def golden_biosnap_slice():
    """First 8 BioSNAP relations through the full TIP-cat encoder with tip.py:14 dims."""
    from tip_amd.data import build_data_dict
    from oracle.tip_oracle import init_params
    d = build_data_dict(max_relations=8)
    p = init_params(d['n_drug'], d['n_prot'], 8, seed=7)
    enc = ref.FMEncoder('cpu', d['n_drug'], 8, d['n_prot'], d['n_prot'], d['n_drug'],
                        prot_drug_dim=16, num_base=32, n_embed=48, n_hid1=32, n_hid2=16, mod='cat')
    sd = enc.state_dict()
    for k in sd:
        sd[k] = p[k].clone()
    enc.load_state_dict(sd)
    torch.manual_seed(3)
    up = torch.randn(d['n_drug'], 16)
    z = enc(d['d_feat'], d['dd_train_idx'], d['dd_train_et'], d['dd_train_range'], d['d_norm'],
            d['p_feat'], d['pp_train_indices'], d['dp_edge_index'], d['dp_range_list'])
    (z * up).sum().backward()
    gr = grads_of(enc)
    gw1 = gr.pop('grad.pp_encoder.conv1.lin.weight')
    npz('biosnap_slice8', z=z, upstream=up, param_seed=7, n_edges=d['dd_train_idx'].shape[1],
        edge_checksum=int(d['dd_train_idx'].sum()), **gr,
        **{'grad.pp_encoder.conv1.lin.weight[:, ::16]': gw1[:, ::16],
           'grad.pp_encoder.conv1.lin.weight.rowsum': gw1.sum(1)})


def golden_drug_features(seed):
    """the reference's FMEncoder fed the REAL sparse drug features
    ([I | mono side effects], 645 x 10 829, data/utils.py:117-132) and a non-unit d_norm, cat and add.
    Also pins tip_amd.data.mono_drug_features against the reference's own loader."""
    from tip_amd.data import mono_drug_features
    d_feat, d_norm = mono_drug_features()
    if not ONLY or 'check_loader' in ONLY:
        cwd = os.getcwd()
        os.chdir('/root/reference')
        try:
            from data.utils import load_data_torch
            et = pickle.load(open('./data/decagon_et.pkl', 'rb'))
            ref_feat = load_data_torch('./data/', et, mono=True)['d_feat'].coalesce()
        finally:
            os.chdir(cwd)
        assert ref_feat.shape == d_feat.shape and torch.equal(ref_feat.indices(), d_feat.indices()) \
            and torch.equal(ref_feat.values(), d_feat.values()), 'mono_drug_features != load_data_torch(mono=True)'
        print('d_feat identical to the reference loader: %s, %d entries' % (tuple(d_feat.shape), d_feat._nnz()))
    n_drug, n_feat = d_feat.shape
    g = small_graph(seed, n_drug=n_drug, n_prot=211, sizes=(400, 1, 60, 900, 30, 170, 2500))
    for mod in ('cat', 'add'):
        torch.manual_seed(seed)
        kw = dict(prot_drug_dim=8, num_base=5, n_embed=12 if mod == 'cat' else 8, n_hid1=8, n_hid2=4)
        enc = ref.FMEncoder('cpu', n_feat, g['n_rel'], g['n_prot'], g['n_prot'], n_drug, mod=mod, **kw)
        randomize_(enc, seed)
        up = torch.randn(n_drug, 4)
        z = enc(d_feat, g['dd_idx'], g['dd_et'], g['dd_range'], d_norm, ref_sparse_id(g['n_prot']), g['pp_idx'],
                g['dp_idx'], None)
        (z * up).sum().backward()
        gg = {k: v for k, v in g.items() if k != 'd_norm'}
        npz('encoder_mono_%s' % mod, z=z, upstream=up, mod=mod, n_feat=n_feat, **gg, **params_of(enc), **grads_of(enc),
            **{'cfg.' + k: v for k, v in kw.items()})


ONLY = [a for a in sys.argv[1:]]

if __name__ == '__main__' and ONLY:
    os.makedirs(OUT, exist_ok=True)
    if 'drug_features' in ONLY:
        golden_drug_features(23)
    if 'fast_route' in ONLY:
        golden_rgcn_fast(31, True, 'rgcn_fast_sym')
        golden_rgcn_fast(32, False, 'rgcn_fast_directed')
        golden_encoder_fast(33, 'cat', True, 'encoder_fast_cat_sym')
        golden_encoder_fast(34, 'add', True, 'encoder_fast_add_sym')
        golden_encoder_fast(35, 'cat', False, 'encoder_fast_cat_directed')
    sys.exit(0)

if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    golden_rgcn(11, 10, 6, 5, True, 'rgcn_sym')
    golden_rgcn(12, 7, 3, 2, False, 'rgcn_directed')
    golden_hier(13)
    golden_pp(14)
    golden_decoder(15)
    golden_nn_decoder(19)
    golden_encoder(16, 'cat', 'encoder_cat_small')
    golden_encoder(17, 'add', 'encoder_add_small')
    golden_tip(18)
    golden_biosnap_slice()
    golden_drug_features(23)
    golden_rgcn_fast(31, True, 'rgcn_fast_sym')
    golden_rgcn_fast(32, False, 'rgcn_fast_directed')
    golden_encoder_fast(33, 'cat', True, 'encoder_fast_cat_sym')
    golden_encoder_fast(34, 'add', True, 'encoder_fast_add_sym')
    golden_encoder_fast(35, 'cat', False, 'encoder_fast_cat_directed')
