"""The oracle (CPU restatement) against the golden vectors recorded from the reference's own code.

This is what pins `oracle/tip_oracle.py`: every fixture under tests/golden was produced by
`oracle/make_golden.py` running /root/reference/src/layers.py unchanged (autograd gradients).
"""
import numpy as np
import pytest
import torch

from oracle import tip_oracle as O
from conftest import load_golden

TOL = dict(rtol=2e-4, atol=2e-5)


def close(a, b, **kw):
    tol = dict(TOL)
    tol.update(kw)
    torch.testing.assert_close(a, b.to(a.dtype), **tol)


@pytest.mark.parametrize('name', ['rgcn_sym', 'rgcn_directed'])
@pytest.mark.parametrize('dtype', [torch.float32, torch.float64])
def test_rgcn_layer(name, dtype):
    g = load_golden(name, dtype)
    out, saved = O.rgcn_fwd(g['x'], g['dd_idx'], g['dd_range'], g['basis'], g['att'], g['root'])
    close(out, g['out'])
    close(out, g['out_bmm_variant'])
    gx, gb, ga, gr = O.rgcn_bwd(g['upstream'], g['x'], g['dd_idx'], g['basis'], g['att'], g['root'], saved)
    close(gx, g['grad_x'])
    close(gb, g['grad.basis'])
    close(ga, g['grad.att'])
    close(gr, g['grad.root'])
    ref_shaped = O.rgcn_fwd_reference_shaped(g['x'], g['dd_idx'], g['dd_range'], g['basis'], g['att'], g['root'])
    close(ref_shaped, g['out'])


@pytest.mark.parametrize('name', ['rgcn_fast_sym', 'rgcn_fast_directed'])
def test_rgcn_two_layers_at_reference_dims(name):
    """Round 6: 64 -> 32 -> 16, num_base 32 (tip.py:14) on a 61-drug graph with self pairs, duplicate edges, one- and two-edge
    relations and isolated drugs -- recorded from the reference's MyRGCNConv2 x 2 with the ReLU of src/layers.py:547 between."""
    g = load_golden(name)
    ei, rg = g['dd_idx'], g['dd_range']
    assert bool((ei[0] == ei[1]).any()) and int((rg[:, 1] - rg[:, 0]).min()) <= 2
    l1 = [g['l1.' + k] for k in ('basis', 'att', 'root')]
    l2 = [g['l2.' + k] for k in ('basis', 'att', 'root')]
    h, s1 = O.rgcn_fwd(g['x'], ei, rg, *l1)
    close(h, g['hidden'], rtol=1e-5, atol=1e-5)
    x1 = torch.relu(h)
    out, s2 = O.rgcn_fwd(x1, ei, rg, *l2)
    close(out, g['out'], rtol=1e-5, atol=1e-5)
    gx1, gb2, ga2, gr2 = O.rgcn_bwd(g['upstream'], x1, ei, *l2, s2)
    gh = gx1 * (h > 0).to(gx1.dtype)
    close(gh, g['grad_hidden'], rtol=1e-4, atol=1e-5)
    gx, gb1, ga1, gr1 = O.rgcn_bwd(gh, g['x'], ei, *l1, s1)
    for got, key in ((gx, 'grad_x'), (gb1, 'grad.l1.basis'), (ga1, 'grad.l1.att'), (gr1, 'grad.l1.root'),
                     (gb2, 'grad.l2.basis'), (ga2, 'grad.l2.att'), (gr2, 'grad.l2.root')):
        close(got, g[key], rtol=1e-4, atol=1e-5 * float(g[key].abs().max()))


def test_hier_conv():
    g = load_golden('hier_conv')
    out, saved = O.hier_conv_fwd(g['x'], g['dp_idx'], g['weight'], g['n_source'])
    close(out, g['out'])
    gx, gw = O.hier_conv_bwd(g['upstream'], g['x'].shape[0], g['dp_idx'], g['weight'], g['n_source'], saved)
    close(gx, g['grad_x'])
    close(gw, g['grad.weight'])


def test_pp_encoder_identity_and_dense():
    g = load_golden('pp_encoder')
    args = (g['conv1.lin.weight'], g['conv1.bias'], g['conv2.lin.weight'], g['conv2.bias'])
    out, saved = O.pp_encoder_fwd(*args, g['pp_idx'], g['n_prot'])
    close(out, g['out'])
    gw1, gb1, gw2, gb2 = O.pp_encoder_bwd(g['upstream'], args[0], args[2], saved)
    close(gw1, g['grad.conv1.lin.weight'])
    close(gb1, g['grad.conv1.bias'])
    close(gw2, g['grad.conv2.lin.weight'])
    close(gb2, g['grad.conv2.bias'])

    g = load_golden('pp_encoder_dense')
    args = (g['conv1.lin.weight'], g['conv1.bias'], g['conv2.lin.weight'], g['conv2.bias'])
    out, saved = O.pp_encoder_fwd(*args, g['pp_idx'], g['x'].shape[0], x=g['x'])
    close(out, g['out'])
    gw1, gb1, gw2, gb2 = O.pp_encoder_bwd(g['upstream'], args[0], args[2], saved)
    close(gw1, g['grad.conv1.lin.weight'])


def test_gcn_norm_self_loops_and_isolated():
    g = load_golden('pp_encoder')
    row, col, norm = O.gcn_norm(g['pp_idx'], g['n_prot'])
    n = g['n_prot']
    # exactly one loop per node, the pre-existing loops (3,3),(9,9),(9,9) were replaced
    assert int((row == col).sum()) == n
    deg = torch.bincount(col, minlength=n)
    assert int(deg.min()) == 1                      # isolated proteins keep only their loop
    close(norm, (deg[row].float() * deg[col].float()).rsqrt())


def test_decoder_and_loss():
    g = load_golden('decoder')
    for sig in (True, False):
        s = O.distmult_fwd(g['z'], g['dd_idx'], g['dd_et'], g['weight'], sigmoid=sig)
        close(s, g['score_%d' % sig])
        gz, gw = O.distmult_bwd(g['upstream'], g['z'], g['dd_idx'], g['dd_et'], g['weight'], sigmoid=sig)
        close(gz, g['grad_z_%d' % sig])
        close(gw, g['grad_weight_%d' % sig])
    pos = O.distmult_fwd(g['z'], g['dd_idx'], g['dd_et'], g['weight'])
    neg = O.distmult_fwd(g['z'], g['neg_idx'], g['dd_et'], g['weight'])
    close(O.tip_loss(pos, neg), g['loss'])
    gp, gn = O.tip_loss_bwd(pos, neg)
    gz1, gw1 = O.distmult_bwd(gp, g['z'], g['dd_idx'], g['dd_et'], g['weight'])
    gz2, gw2 = O.distmult_bwd(gn, g['z'], g['neg_idx'], g['dd_et'], g['weight'])
    close(gz1 + gz2, g['loss_grad_z'])
    close(gw1 + gw2, g['loss_grad_weight'])


def _encoder_inputs(g):
    p = {k: v for k, v in g.items() if isinstance(v, torch.Tensor) and v.is_floating_point()
         and not k.startswith('grad.') and k not in ('z', 'upstream', 'd_norm')}
    data = dict(dd_train_idx=g['dd_idx'], dd_train_range=g['dd_range'], d_norm=g['d_norm'],
                pp_train_indices=g['pp_idx'], dp_edge_index=g['dp_idx'],
                n_drug=g['n_drug'], n_prot=g['n_prot'])
    return p, data


@pytest.mark.parametrize('name', ['encoder_cat_small', 'encoder_add_small', 'encoder_fast_cat_sym', 'encoder_fast_add_sym',
                                  'encoder_fast_cat_directed'])
def test_fm_encoder(name):
    g = load_golden(name)
    p, data = _encoder_inputs(g)
    mod = str(g['mod'])
    z, saved = O.fm_encoder_fwd(p, data, mod)
    close(z, g['z'])
    grads = O.fm_encoder_bwd(g['upstream'], p, data, saved, mod)
    assert set(grads) == set(p)
    for k, v in grads.items():
        close(v, g['grad.' + k])


@pytest.mark.parametrize('mod', ['cat', 'add'])
def test_fm_encoder_real_drug_features(mod):
    """the oracle with the sparse [I | mono] drug features and the non-unit d_norm
    against the reference's FMEncoder on the same inputs (fixture from oracle/make_golden.py)."""
    from tip_amd.data import mono_drug_features
    g = load_golden('encoder_mono_' + mod)
    d_feat, d_norm = mono_drug_features()
    assert d_feat.shape == (g['n_drug'], int(g['n_feat'])) and float(d_norm.max()) > 100
    data = dict(d_feat=d_feat, d_norm=d_norm, dd_train_idx=g['dd_idx'], dd_train_range=g['dd_range'],
                pp_train_indices=g['pp_idx'], dp_edge_index=g['dp_idx'], n_drug=g['n_drug'], n_prot=g['n_prot'])
    p = {k[len('grad.'):]: g[k[len('grad.'):]] for k in g if k.startswith('grad.')}
    z, saved = O.fm_encoder_fwd(p, data, mod)
    close(z, g['z'])
    grads = O.fm_encoder_bwd(g['upstream'], p, data, saved, mod)
    assert set(grads) == set(p)
    for k, v in grads.items():
        close(v, g['grad.' + k])


def test_biosnap_slice_encoder():
    from tip_amd.data import build_data_dict
    g = load_golden('biosnap_slice8')
    d = build_data_dict(max_relations=8)
    assert d['dd_train_idx'].shape[1] == g['n_edges'] and int(d['dd_train_idx'].sum()) == g['edge_checksum']
    p = O.init_params(d['n_drug'], d['n_prot'], 8, seed=g['param_seed'])
    z, saved = O.fm_encoder_fwd(p, d, 'cat')
    close(z, g['z'], rtol=1e-4, atol=1e-5)
    grads = O.fm_encoder_bwd(g['upstream'], p, d, saved, 'cat')
    for k, v in grads.items():
        if k == 'pp_encoder.conv1.lin.weight':
            close(v[:, ::16], g['grad.' + k + '[:, ::16]'], rtol=2e-4, atol=2e-5 * float(g['grad.' + k + '[:, ::16]'].abs().max()))
            close(v.sum(1), g['grad.' + k + '.rowsum'], rtol=2e-4, atol=2e-5 * float(g['grad.' + k + '.rowsum'].abs().max()))
        else:
            close(v, g['grad.' + k], rtol=2e-4, atol=2e-5 * float(g['grad.' + k].abs().max()))


def test_tip_end_to_end_loss_and_metrics():
    g = load_golden('tip_add_small')
    p = {k[len('encoder.'):]: v for k, v in g.items() if isinstance(k, str) and k.startswith('encoder.')}
    data = dict(dd_train_idx=g['dd_train_idx'], dd_train_range=g['dd_train_range'], d_norm=g['d_norm'],
                pp_train_indices=g['pp_train_indices'], dp_edge_index=g['dp_edge_index'],
                n_drug=g['n_drug'], n_prot=g['n_prot'])
    z, saved = O.fm_encoder_fwd(p, data, 'add')
    close(z, g['embeddings'])
    w = g['decoder.weight']
    pos = O.distmult_fwd(z, g['dd_train_idx'], g['dd_train_et'], w)
    neg = O.distmult_fwd(z, g['train_neg'], g['dd_train_et'], w)
    close(O.tip_loss(pos, neg), g['loss'])
    gp, gn = O.tip_loss_bwd(pos, neg)
    gz1, gw1 = O.distmult_bwd(gp, z, g['dd_train_idx'], g['dd_train_et'], w)
    gz2, gw2 = O.distmult_bwd(gn, z, g['train_neg'], g['dd_train_et'], w)
    close(gw1 + gw2, g['grad.decoder.weight'])
    grads = O.fm_encoder_bwd(gz1 + gz2, p, data, saved, 'add')
    for k, v in grads.items():
        close(v, g['grad.encoder.' + k], rtol=2e-4, atol=2e-5 * float(g['grad.encoder.' + k].abs().max()))
    # test(): per-relation metrics on the recorded fixed test negatives
    ps = O.distmult_fwd(z, g['dd_test_idx'], g['dd_test_et'], w)
    ns = O.distmult_fwd(z, g['test_neg'], g['dd_test_et'], w)
    for i, (a, b) in enumerate(g['dd_test_range'].tolist()):
        y = np.r_[np.ones(b - a), np.zeros(b - a)]
        s = np.r_[ps[a:b].numpy(), ns[a:b].numpy()]
        np.testing.assert_allclose(O.auprc_auroc_ap(y, s), g['record'][:, i].numpy(), rtol=1e-6)


def test_negative_sampling_restatement_distribution():
    """A7 has no sample-level oracle (global numpy RNG); check the documented properties."""
    g = load_golden('tip_add_small')
    rng = np.random.RandomState(5)
    n = g['n_drug']
    neg = O.typed_negative_sampling(g['dd_train_idx'], n, g['dd_train_range'], rng)
    assert neg.shape == g['dd_train_idx'].shape and neg.dtype == torch.int64
    assert int(neg.min()) >= 0 and int(neg.max()) < n


def test_nn_decoder_restatement():
    g = load_golden('nn_decoder')
    s = O.nn_decoder_fwd(g['z'], g['dd_idx'], g['dd_et'], g['w1_l1'], g['w1_l2'], g['w2_l1'], g['w2_l2'])
    close(s, g['score'])
    # the explicit backward against the reference's autograd gradients (upstream = the golden's weights of the scores)
    gs = O.nn_decoder_bwd(g['upstream'], g['z'], g['dd_idx'], g['dd_et'], g['w1_l1'], g['w1_l2'], g['w2_l1'], g['w2_l2'], chunk=7)
    close(gs[0], g['grad_z'], atol=1e-6)
    for got, k in zip(gs[1:], ('w1_l1', 'w1_l2', 'w2_l1', 'w2_l2')):
        close(got, g['grad.' + k], atol=1e-6)
