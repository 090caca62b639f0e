"""-m gpu: the `tip_amd.layers` modules (reference `forward()` signatures) against the golden
vectors recorded from the reference's own code, and against the oracle at BioSNAP scale."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import tip_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def close(got, want, rtol=2e-4, atol=None):
    want = want.detach().to('cpu', torch.float64)
    got = got.detach().to('cpu', torch.float64)
    if atol is None:
        atol = 2e-5 * max(1.0, float(want.abs().max()))
    torch.testing.assert_close(got, want, rtol=rtol, atol=atol)


# Tolerances of the module-level comparisons (round 6, calibrated with TIPK_ERRLOG, tests/conftest.py): the largest error any
# of these sites showed on MI355X was 3e-7 max|want| for activations and 1.1e-6 max|want| for gradients (4e-6 for the NNDecoder
# tables at BioSNAP size), both sides fp32 with different summation orders.  The bounds below are 10-30 x that.
def close_act(got, want):
    """activations / embeddings: |d| <= 1e-5 |want| + 1e-5 max|want|."""
    close(got, want, rtol=1e-5, atol=1e-5 * float(want.detach().abs().max()) + 1e-12)


def close_grad(got, want, scale=1.0):
    """gradients: |d| <= 1e-4 |want| + 2e-5 max|want|  (scale: the 64-bit fixed-point decoder sums at full size)."""
    close(got, want, rtol=1e-4 * scale, atol=2e-5 * scale * float(want.detach().abs().max()) + 1e-12)


def load_params(module, g, prefix=''):
    sd = module.state_dict()
    for k in sd:
        sd[k] = g[prefix + k].clone()
    module.load_state_dict(sd)                        # reference state_dict names load unchanged
    return module.to(DEV)


@pytest.mark.parametrize('name', ['rgcn_sym', 'rgcn_directed'])
def test_rgcn_layers(name):
    from tip_amd.layers import MyRGCNConv, MyRGCNConv2
    g = load_golden(name)
    nb, d_in, d_out = g['basis'].shape
    r = g['att'].shape[0]
    for cls in (MyRGCNConv2, MyRGCNConv):
        m = load_params(cls(d_in, d_out, r, nb, after_relu=False), g)
        x = g['x'].to(DEV).requires_grad_(True)
        ei, et, rg = g['dd_idx'].to(DEV), g['dd_et'].to(DEV), g['dd_range'].to(DEV)
        out = m(x, ei, et, rg) if cls is MyRGCNConv2 else m(x, ei, et)
        close(out, g['out'])
        (out * g['upstream'].to(DEV)).sum().backward()
        close(x.grad, g['grad_x'])
        for k in ('basis', 'att', 'root'):
            close(getattr(m, k).grad, g['grad.' + k])
        out2 = m(x, ei, et, rg) if cls is MyRGCNConv2 else m(x, ei, et)     # cached plans, same bits
        assert torch.equal(out, out2)


def _launch_labels(fn):
    """Labels of the launches `fn` makes (ops' per-launch timing hooks), as one string."""
    from tip_amd import ops
    ops.timing_start()
    try:
        fn()
        torch.cuda.synchronize()
    finally:
        rec = ops.timing_stop()
    return ' '.join(sorted(rec))


@pytest.mark.parametrize('name', ['rgcn_fast_sym', 'rgcn_fast_directed'])
def test_rgcn_pair_route_against_reference_golden(name):
    """Round 6 (VERDICT r5 missing #4): the PAIR-FORM route (n_bases 32, 64 -> 32 -> 16) against outputs and autograd gradients of
    the reference's own MyRGCNConv2 x 2 on a graph with self pairs, duplicate edges inside a relation, one- and two-edge
    relations and isolated drugs -- and the test asserts that this route, not the generic one, produced the numbers."""
    from tip_amd.layers import MyRGCNConv2
    g = load_golden(name)
    r = g['l1.att'].shape[0]
    m1 = load_params(MyRGCNConv2(64, 32, r, 32, after_relu=False), g, 'l1.')
    m2 = load_params(MyRGCNConv2(32, 16, r, 32, after_relu=True), g, 'l2.')
    ei, et, rg = g['dd_idx'].to(DEV), g['dd_et'].to(DEV), g['dd_range'].to(DEV)
    x = g['x'].to(DEV).requires_grad_(True)
    box = {}

    def run():
        # exactly as FMEncoder.forward drives the two layers (ReLU fused, slab sum handed over, cells of both layers together)
        tok = object()
        h = m1(x, ei, et, rg, fuse_relu='gated_downstream', defer_output=True, next_layer=m2, cells_token=tok)
        box['out'] = m2(h, ei, et, rg, gate_input=True, cells_token=tok)
        (box['out'] * g['upstream'].to(DEV)).sum().backward()
    labels = _launch_labels(run)
    for need in ('pair_cells', 'pair_grads', 'pair_att_gather', 'sum_slabs_xb'):
        assert need in labels, (need, labels)
    assert 'gather_sum[dd' not in labels and 'rel_gather' not in labels and 'rel_stream' not in labels, labels
    assert 'pair_cells2' in labels and 'pair_cells[' not in labels, labels      # ONE cell launch for both layers (token handed over)
    graph = m1.graph_for(x.shape[0], ei, rg)
    assert graph.pair_fwd is not None and graph.pair_fwd.symmetric == ('sym' in name)
    close(box['out'], g['out'], rtol=1e-5, atol=1e-5 * float(g['out'].abs().max()))
    close(x.grad, g['grad_x'], rtol=1e-4, atol=1e-5 * float(g['grad_x'].abs().max()))
    for tag, m in (('l1', m1), ('l2', m2)):
        for k in ('basis', 'att', 'root'):
            want = g['grad.%s.%s' % (tag, k)]
            close(getattr(m, k).grad, want, rtol=1e-4, atol=1e-5 * float(want.abs().max()))
    # the layers one by one (no hand-over, ReLU outside): the hidden activations and their gradient too
    m1.zero_grad(); m2.zero_grad()
    x2 = g['x'].to(DEV).requires_grad_(True)
    h = m1(x2, ei, et, rg)
    h.retain_grad()
    out = m2(torch.relu(h), ei, et, rg)
    (out * g['upstream'].to(DEV)).sum().backward()
    close(h, g['hidden'], rtol=1e-5, atol=1e-5 * float(g['hidden'].abs().max()))
    close(h.grad, g['grad_hidden'], rtol=1e-4, atol=1e-5 * float(g['grad_hidden'].abs().max()))
    close(out, g['out'], rtol=1e-5, atol=1e-5 * float(g['out'].abs().max()))
    close(x2.grad, g['grad_x'], rtol=1e-4, atol=1e-5 * float(g['grad_x'].abs().max()))


@pytest.mark.parametrize('name', ['rgcn_fast_sym', 'rgcn_fast_directed'])
def test_graph_handle_pair_form_is_the_modules_pair_form_bit_for_bit(name):
    """The op-level handle after `tipk_graph_prepare_rgcn` (plans built in C++, include/tipk.h section 10c) against the PyTorch
    module on the same layer: the same plans, the same launches -- the same bits, forward and all four gradients."""
    import ctypes as C
    from tip_amd import _lib
    from tip_amd.layers import MyRGCNConv2
    L = _lib.lib()
    g = load_golden(name)
    r = g['l1.att'].shape[0]
    m1 = load_params(MyRGCNConv2(64, 32, r, 32, after_relu=False), g, 'l1.')
    ei, et, rg = g['dd_idx'].to(DEV), g['dd_et'].to(DEV), g['dd_range'].to(DEV)
    x = g['x'].to(DEV).requires_grad_(True)
    up = torch.randn(x.shape[0], 32, generator=torch.Generator().manual_seed(1)).to(DEV)
    want = m1(x, ei, et, rg)
    assert m1.graph_for(x.shape[0], ei, rg).pair_fwd is not None
    want.backward(up)
    h = C.c_void_p()
    ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    assert L.tipk_graph_build(ptr(ei), None, ptr(rg), 8, ei.shape[1], x.shape[0], r, None, C.byref(h)) == 0
    try:
        assert L.tipk_graph_rgcn_route(h, 32, 32) == 0
        assert L.tipk_graph_prepare_rgcn(h, 32, 32) == 0
        assert L.tipk_graph_rgcn_route(h, 32, 32) == 2
        assert L.tipk_graph_prepare_rgcn(h, 5, 32) == -2 and L.tipk_graph_rgcn_route(h, 5, 32) == 0       # not a pair-form shape
        ws = torch.empty(L.tipk_rgcn_workspace_bytes(h, 64, 32, 32), dtype=torch.uint8, device=DEV)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        xd, basis, att, root = x.detach(), m1.basis.detach().contiguous(), m1.att.detach().contiguous(), m1.root.detach().contiguous()
        out = torch.empty(x.shape[0], 32, device=DEV)
        assert L.tipk_rgcn_fwd(h, ptr(xd), 64, 64, ptr(basis), ptr(att), ptr(root), 32, 32, 0, ptr(out), 32, ptr(ws), ws.numel(), st) == 0
        assert torch.equal(out, want.detach())
        gx, gb, ga, gr = torch.empty_like(xd), torch.empty_like(basis), torch.empty_like(att), torch.empty_like(root)
        assert L.tipk_rgcn_bwd_ex(h, ptr(xd), 64, 64, ptr(basis), ptr(att), ptr(root), 32, 32, ptr(up), 32, None, 0, ptr(gx), 64, ptr(gb),
                                  ptr(ga), ptr(gr), ptr(ws), ws.numel(), 1, st) == 0
        for got, ref in ((gx, x.grad), (gb, m1.basis.grad), (ga, m1.att.grad), (gr, m1.root.grad)):
            assert torch.equal(got, ref)
        # ... and without TIPK_RGCN_WORKSPACE_FROM_FWD, on a workspace that holds nothing (NaN bit patterns): cells and XB recomputed
        ws.fill_(255)
        cold = [torch.empty_like(t) for t in (gx, gb, ga, gr)]
        assert L.tipk_rgcn_bwd(h, ptr(xd), 64, 64, ptr(basis), ptr(att), ptr(root), 32, 32, ptr(up), 32, None, 0, ptr(cold[0]), 64,
                               ptr(cold[1]), ptr(cold[2]), ptr(cold[3]), ptr(ws), ws.numel(), st) == 0
        for got, ref in zip(cold, (gx, gb, ga, gr)):
            assert torch.equal(got, ref)
        # the ReLU inside the layer: forward with relu = 1, the mask applied by the backward pass from the output
        out_r = torch.empty_like(out)
        assert L.tipk_rgcn_fwd(h, ptr(xd), 64, 64, ptr(basis), ptr(att), ptr(root), 32, 32, 1, ptr(out_r), 32, ptr(ws), ws.numel(), st) == 0
        assert torch.equal(out_r, torch.relu(out))
        assert L.tipk_rgcn_bwd_ex(h, ptr(xd), 64, 64, ptr(basis), ptr(att), ptr(root), 32, 32, ptr(up), 32, ptr(out_r), 32, ptr(cold[0]), 64,
                                  ptr(cold[1]), ptr(cold[2]), ptr(cold[3]), ptr(ws), ws.numel(), 1, st) == 0
        x3 = g['x'].to(DEV).requires_grad_(True)
        m1.zero_grad()
        torch.relu(m1(x3, ei, et, rg)).backward(up)
        for got, ref in zip(cold, (x3.grad, m1.basis.grad, m1.att.grad, m1.root.grad)):
            close(got, ref, rtol=1e-5, atol=1e-6 * float(ref.abs().max()))
        assert L.tipk_graph_release_host(h) == 0
        assert L.tipk_graph_prepare_rgcn(h, 32, 32) == 0                   # prepared already: nothing to build
        assert L.tipk_graph_prepare_rgcn(h, 16, 32) == -1                  # a new shape needs the edge list back
    finally:
        assert L.tipk_graph_destroy(h) == 0


@pytest.mark.parametrize('name', ['encoder_fast_cat_sym', 'encoder_fast_add_sym', 'encoder_fast_cat_directed'])
def test_fm_encoder_fast_route_against_reference_golden(name):
    """FMEncoder at the dims of tip.py:14 / :17 against the reference's own forward + autograd on the nasty 61-drug graph: the
    route the bench times (fused P -> D + mix, layer hand-over, pair cells of both layers in one launch, pair-form backward)."""
    from tip_amd.layers import FMEncoder
    from tip_amd.utils import sparse_id
    g = load_golden(name)
    mod = str(g['mod'])
    cfg = {k[4:]: int(v) for k, v in g.items() if isinstance(k, str) and k.startswith('cfg.')}
    enc = load_params(FMEncoder(DEV, g['n_drug'], g['n_rel'], g['n_prot'], g['n_prot'], g['n_drug'], mod=mod, **cfg), g)
    args = (sparse_id(g['n_drug']).to(DEV), g['dd_idx'].to(DEV), g['dd_et'].to(DEV), g['dd_range'].to(DEV),
            g['d_norm'].to(DEV), sparse_id(g['n_prot']).to(DEV), g['pp_idx'].to(DEV), g['dp_idx'].to(DEV), None)
    box = {}

    def run():
        box['z'] = enc(*args)
        (box['z'] * g['upstream'].to(DEV)).sum().backward()
    labels = _launch_labels(run)
    for need in ('pair_cells2', 'pair_grads', 'pair_att_gather2', 'sum_slabs_xb', 'drug_mix_gather_xb_fwd', 'pd_stage_bwd'):
        assert need in labels, (need, labels)
    close(box['z'], g['z'], rtol=1e-5, atol=1e-5 * float(g['z'].abs().max()))
    for k, p in enc.named_parameters():
        want = g['grad.' + k]
        close(p.grad, want, rtol=1e-4, atol=1e-5 * float(want.abs().max()))
    # a second pass on the cached plans: the same bits
    enc.zero_grad()
    z2 = enc(*args)
    assert torch.equal(z2, box['z'])


def test_encoder_step_is_sixteen_launches_and_matches_the_per_layer_nodes(monkeypatch):
    """Round 6: FMEncoder.forward + backward as ONE autograd node (tip_amd/encoder.py) -- 8 + 8 launches on the reference's
    dims -- against the same pass on the per-layer nodes (TIPK_NO_ENCODER_STEP=1: 9 + 11 launches); a backward pass that runs
    after ANOTHER forward pass recomputes the graph's buffers."""
    from tip_amd import ops
    from tip_amd.layers import FMEncoder
    from tip_amd.utils import sparse_id
    g = load_golden('encoder_fast_cat_sym')
    cfg = {k[4:]: int(v) for k, v in g.items() if isinstance(k, str) and k.startswith('cfg.')}
    args = (sparse_id(g['n_drug']).to(DEV), g['dd_idx'].to(DEV), g['dd_et'].to(DEV), g['dd_range'].to(DEV),
            g['d_norm'].to(DEV), sparse_id(g['n_prot']).to(DEV), g['pp_idx'].to(DEV), g['dp_idx'].to(DEV), None)
    up = g['upstream'].to(DEV)

    def run(per_layer):
        if per_layer:
            monkeypatch.setenv('TIPK_NO_ENCODER_STEP', '1')
        else:
            monkeypatch.delenv('TIPK_NO_ENCODER_STEP', raising=False)
        enc = load_params(FMEncoder(DEV, g['n_drug'], g['n_rel'], g['n_prot'], g['n_prot'], g['n_drug'], mod='cat', **cfg), g)
        ops.timing_start()
        z = enc(*args)
        (z * up).sum().backward()
        torch.cuda.synchronize()
        rec = ops.timing_stop()
        return enc, z, sum(v[0] for v in rec.values()), rec
    e_new, z_new, n_new, rec_new = run(False)
    e_old, z_old, n_old, _ = run(True)
    assert n_new == 16 and n_old == 20, (n_new, n_old, sorted(rec_new))
    close_act(z_new, z_old.cpu())
    for (k, a), (_, b) in zip(e_new.named_parameters(), e_old.named_parameters()):
        assert a.grad.shape == b.grad.shape and a.grad.stride() == b.grad.stride(), k
        close_grad(a.grad, b.grad.cpu())
    # two forward passes, then the FIRST one's backward: its cells / XB were overwritten -- recomputed, same gradients
    monkeypatch.delenv('TIPK_NO_ENCODER_STEP', raising=False)
    e2 = load_params(FMEncoder(DEV, g['n_drug'], g['n_rel'], g['n_prot'], g['n_prot'], g['n_drug'], mod='cat', **cfg), g)
    z_a = e2(*args)
    z_b = e2(*args)                                                      # bumps the stamps of both graphs' buffers ...
    n, nb = g['n_drug'], cfg['num_base']
    for layer in (e2.rgcn1, e2.rgcn2):                                   # ... whose contents are then lost
        cells, xb, _ = layer._cache.value.pair_buffers(n, nb, layer.out_channels, torch.device(DEV))
        xb[:n].fill_(7.0)
        cells.view(-1, nb)[:n * n].mul_(3.0)
    (z_a * up).sum().backward()
    for (k, a), (_, b) in zip(e2.named_parameters(), e_new.named_parameters()):
        close_grad(a.grad, b.grad.cpu())


def test_hierarchy_conv():
    from tip_amd.layers import MyHierarchyConv
    g = load_golden('hier_conv')
    n_src = g['n_source']
    m = load_params(MyHierarchyConv(16, 8, n_src, g['x'].shape[0] - n_src), g)
    x = g['x'].to(DEV).requires_grad_(True)
    out = m(x, g['dp_idx'].to(DEV), None)
    close(out, g['out'])
    (out * g['upstream'].to(DEV)).sum().backward()
    close(x.grad, g['grad_x'])
    close(m.weight.grad, g['grad.weight'])


def test_pp_encoder_identity_sparse_dense_features():
    from tip_amd.layers import PPEncoder
    from tip_amd.utils import sparse_id
    g = load_golden('pp_encoder')
    n = g['n_prot']
    m = load_params(PPEncoder(n), g)
    out = m(sparse_id(n).to(DEV), g['pp_idx'].to(DEV))
    close(out, g['out'])
    (out * g['upstream'].to(DEV)).sum().backward()
    for k in ('conv1.lin.weight', 'conv1.bias', 'conv2.lin.weight', 'conv2.bias'):
        mod, attr = k.rsplit('.', 1)
        close(getattr(m.get_submodule(mod), attr).grad, g['grad.' + k])
    # a general sparse feature matrix (identity with shuffled storage + explicit values) and the
    # dense path give the same numbers through the plan-based SpMM / GEMM
    perm = torch.randperm(n)
    sp = torch.sparse_coo_tensor(torch.stack([perm, perm]), torch.ones(n), (n, n)).to(DEV)
    sp._tipk_identity = False
    m2 = load_params(PPEncoder(n), g)
    close(m2(sp, g['pp_idx'].to(DEV)), g['out'])
    gd = load_golden('pp_encoder_dense')
    m3 = load_params(PPEncoder(24), gd)
    x = gd['x'].to(DEV).requires_grad_(True)
    out3 = m3(x, gd['pp_idx'].to(DEV))
    close(out3, gd['out'])
    (out3 * gd['upstream'].to(DEV)).sum().backward()
    close(x.grad, gd['grad_x'])
    close(m3.conv1.lin.weight.grad, gd['grad.conv1.lin.weight'])


def test_decoder_module_and_loss():
    from tip_amd.layers import MultiInnerProductDecoder
    g = load_golden('decoder')
    m = load_params(MultiInnerProductDecoder(4, g['weight'].shape[0]), g)
    ei, et = g['dd_idx'].to(DEV), g['dd_et'].to(DEV)
    for sig in (True, False):
        z = g['z'].to(DEV).requires_grad_(True)
        m.weight.grad = None
        s = m(z, ei, et, sigmoid=sig)
        close(s, g['score_%d' % sig])
        (s * g['upstream'].to(DEV)).sum().backward()
        close(z.grad, g['grad_z_%d' % sig])
        close(m.weight.grad, g['grad_weight_%d' % sig])
    for fused in (True, False):
        z = g['z'].to(DEV).requires_grad_(True)
        m.weight.grad = None
        neg = g['neg_idx'].to(DEV)
        if fused:
            loss = m.objective(z, ei, neg, et)
        else:
            loss = -torch.log(m(z, ei, et) + 1e-13).mean() - torch.log(1 - m(z, neg, et) + 1e-13).mean()
        close(loss, g['loss'], rtol=2e-5)
        (loss * 1.0).backward()
        close(z.grad, g['loss_grad_z'], atol=1e-6)
        close(m.weight.grad, g['loss_grad_weight'], atol=1e-6)


@pytest.mark.parametrize('name', ['encoder_cat_small', 'encoder_add_small'])
def test_fm_encoder_golden(name):
    from tip_amd.layers import FMEncoder
    from tip_amd.utils import sparse_id
    g = load_golden(name)
    mod = str(g['mod'])
    cfg = {k[4:]: int(v) for k, v in g.items() if isinstance(k, str) and k.startswith('cfg.')}
    enc = FMEncoder(DEV, g['n_drug'], g['n_rel'], g['n_prot'], g['n_prot'], g['n_drug'], mod=mod, **cfg)
    enc = load_params(enc, g)
    z = enc(sparse_id(g['n_drug']).to(DEV), g['dd_idx'].to(DEV), g['dd_et'].to(DEV), g['dd_range'].to(DEV),
            g['d_norm'].to(DEV), sparse_id(g['n_prot']).to(DEV), g['pp_idx'].to(DEV), g['dp_idx'].to(DEV), None)
    close(z, g['z'])
    (z * g['upstream'].to(DEV)).sum().backward()
    for k, p in enc.named_parameters():
        close(p.grad, g['grad.' + k])


@pytest.mark.parametrize('mod', ['cat', 'add'])
def test_fm_encoder_real_drug_features_golden(mod):
    """the CSR SpMM path of `x_drug @ embed` (`_FeatureInput` -> tipk_gather_sum over the
    sparse feature matrix, forward and transposed) with the REAL drug features -- [I | mono side effects],
    645 x 10 829, built as data/utils.py:117-132 builds them -- and a non-unit d_norm, against the
    reference's FMEncoder on the same inputs."""
    from tip_amd.data import mono_drug_features
    from tip_amd.layers import FMEncoder
    from tip_amd.utils import sparse_id
    g = load_golden('encoder_mono_' + mod)
    d_feat, d_norm = mono_drug_features()
    cfg = {k[4:]: int(v) for k, v in g.items() if isinstance(k, str) and k.startswith('cfg.')}
    enc = FMEncoder(DEV, int(g['n_feat']), g['n_rel'], g['n_prot'], g['n_prot'], g['n_drug'], mod=mod, **cfg)
    enc = load_params(enc, g)
    z = enc(d_feat.to(DEV), g['dd_idx'].to(DEV), g['dd_et'].to(DEV), g['dd_range'].to(DEV), d_norm.to(DEV),
            sparse_id(g['n_prot']).to(DEV), g['pp_idx'].to(DEV), g['dp_idx'].to(DEV), None)
    close(z, g['z'])
    (z * g['upstream'].to(DEV)).sum().backward()
    for k, p in enc.named_parameters():
        close(p.grad, g['grad.' + k])
    assert enc._drug_feat._cache.value is not None                     # the sparse-feature plans were used


def test_tip_end_to_end_small():
    """TIP(...) construction, forward() loss with the recorded negatives, backward, test()."""
    from tip_amd.layers import TIP, Setting
    g = load_golden('tip_add_small')
    keys = ['dd_train_idx', 'dd_train_et', 'dd_train_range', 'dd_test_idx', 'dd_test_et', 'dd_test_range',
            'pp_train_indices', 'dp_edge_index', 'dp_range_list', 'd_norm']
    from tip_amd.utils import sparse_id
    d = {k: g[k] for k in keys}
    d.update(n_drug=g['n_drug'], n_prot=g['n_prot'], n_dd_et=g['n_dd_et'], n_drug_feat=g['n_drug'],
             d_feat=sparse_id(g['n_drug']), p_feat=sparse_id(g['n_prot']))
    st = Setting(sp_rate=0.9, lr=0.01, prot_drug_dim=8, n_embed=8, n_hid1=8, n_hid2=4, num_base=3)
    for fused in (True, False):
        model = TIP(st, torch.device(DEV), mod='add', data=d, fused_loss=fused)
        assert model.embeddings.shape == (g['n_drug'], 4) and model.test_neg_index.shape == g['test_neg'].shape
        sd = model.state_dict()
        assert set(sd) == {k for k in g if isinstance(k, str) and (k.startswith('encoder.') or k.startswith('decoder.'))}
        load_params(model, g)
        loss = model(neg_index=g['train_neg'].to(DEV))
        close(loss, g['loss'], rtol=2e-5)
        close(model.embeddings, g['embeddings'])
        loss.backward()
        for k, p in model.named_parameters():
            close_grad(p.grad, g['grad.' + k])
        model.test_neg_index = g['test_neg'].to(DEV)
        rec = model.test(print_output=False)
        np.testing.assert_allclose(rec, g['record'].numpy(), rtol=1e-4, atol=1e-4)
        # default path: negatives drawn on device, loss finite and close to the recorded one
        assert abs(float(model()) - float(g['loss'])) < 0.5
        # serving helper: top-k side effects of drug pairs == a dense re-computation from the embeddings
        pairs = torch.tensor([[0, 1, 2, 3, 5], [1, 0, 4, 3, 2]])
        vals, ids = model.pred_topk(pairs, k=3, max_triples=2 * g['n_dd_et'])          # forces several slices
        z, w = model.embeddings.detach().double().cpu(), model.decoder.weight.detach().double().cpu()
        dense = torch.sigmoid((z[pairs[0]] * z[pairs[1]]) @ w.t())
        want = torch.topk(dense, 3, dim=1)
        close(vals, want.values, rtol=1e-5)
        assert torch.equal(ids.cpu(), want.indices) and vals.shape == (5, 3)
        assert model.pred(pairs[:, :2].to(DEV), torch.tensor([0, 1]).to(DEV)).shape == (2,)


def test_biosnap_slice_against_reference_golden():
    from tip_amd.data import build_data_dict, Data
    from tip_amd.layers import FMEncoder
    g = load_golden('biosnap_slice8')
    dd = build_data_dict(max_relations=8)
    p = O.init_params(dd['n_drug'], dd['n_prot'], 8, seed=g['param_seed'])
    enc = FMEncoder(DEV, dd['n_drug'], 8, dd['n_prot'], dd['n_prot'], dd['n_drug'], prot_drug_dim=16, num_base=32,
                    n_embed=48, n_hid1=32, n_hid2=16, mod='cat')
    enc = load_params(enc, p)
    d = Data.from_dict(dd).to(DEV)
    z = enc(d.d_feat, d.dd_train_idx, d.dd_train_et, d.dd_train_range, d.d_norm, d.p_feat, d.pp_train_indices,
            d.dp_edge_index, d.dp_range_list)
    close_act(z, g['z'])
    (z * g['upstream'].to(DEV)).sum().backward()
    for k, prm in enc.named_parameters():
        if k == 'pp_encoder.conv1.lin.weight':
            close_grad(prm.grad[:, ::16], g['grad.' + k + '[:, ::16]'])
            close_grad(prm.grad.sum(1), g['grad.' + k + '.rowsum'])
        else:
            close_grad(prm.grad, g['grad.' + k])


@pytest.mark.parametrize('dims', [
    dict(prot_drug_dim=32, n_embed=32, n_hid1=64, n_hid2=32, num_base=16, mod='cat'),
    dict(prot_drug_dim=24, n_embed=24, n_hid1=16, n_hid2=8, num_base=5, mod='add'),
    dict(prot_drug_dim=8, n_embed=20, n_hid1=128, n_hid2=4, num_base=32, mod='cat')])
def test_encoder_other_dimensions_vs_oracle(dims):
    """Layer widths other than tip.py's (column-split / non-split LDS kernels, num_base != 32, d_in
    not a power of two): z and every gradient vs the oracle on a 40-relation BioSNAP slice."""
    from tip_amd.data import build_data_dict, Data
    from tip_amd.layers import FMEncoder
    dims = dict(dims)
    mod = dims.pop('mod')
    dd = build_data_dict(max_relations=40)
    R = dd['n_dd_et']
    p = O.init_params(dd['n_drug'], dd['n_prot'], R, mod=mod, seed=7, **dims)
    enc = FMEncoder(DEV, dd['n_drug'], R, dd['n_prot'], dd['n_prot'], dd['n_drug'], mod=mod, **dims)
    enc = load_params(enc, p)
    d = Data.from_dict(dd).to(DEV)
    z = enc(d.d_feat, d.dd_train_idx, d.dd_train_et, d.dd_train_range, d.d_norm, d.p_feat, d.pp_train_indices,
            d.dp_edge_index, d.dp_range_list)
    torch.manual_seed(1)
    up = torch.randn(dd['n_drug'], dims['n_hid2'])
    (z * up.to(DEV)).sum().backward()
    zo, saved = O.fm_encoder_fwd(p, dd, mod)
    go = O.fm_encoder_bwd(up, p, dd, saved, mod)
    close_act(z, zo)
    for k, prm in enc.named_parameters():
        close_grad(prm.grad, go[k])


def test_synthetic_encoder_large_node_set_vs_oracle():
    """More drugs than the LDS-resident kernel takes (N > 1024): the fabric-gather path, its finalize
    launch, dy_products with hundreds of column chunks -- the route BASELINE config 5 takes -- vs the oracle."""
    from tip_amd.data import synthetic_data_dict, Data
    from tip_amd.layers import FMEncoder
    from tip_amd import ops
    dd = synthetic_data_dict(n_drug=1500, n_rel=12, n_edges=60000, seed=5, with_protein_graph=True,
                             n_prot=700, pp_edges=4000, dp_edges=900)
    assert ops.rel_gather_split(1500, 64, False) == 0
    dims = dict(prot_drug_dim=16, n_embed=48, n_hid1=64, n_hid2=32, num_base=32)
    R = dd['n_dd_et']
    p = O.init_params(dd['n_drug'], dd['n_prot'], R, mod='cat', seed=3, **dims)
    enc = FMEncoder(DEV, dd['n_drug'], R, dd['n_prot'], dd['n_prot'], dd['n_drug'], mod='cat', **dims)
    enc = load_params(enc, p)
    d = Data.from_dict(dd).to(DEV)
    z = enc(d.d_feat, d.dd_train_idx, d.dd_train_et, d.dd_train_range, d.d_norm, d.p_feat, d.pp_train_indices,
            d.dp_edge_index, d.dp_range_list)
    torch.manual_seed(2)
    up = torch.randn(dd['n_drug'], dims['n_hid2'])
    (z * up.to(DEV)).sum().backward()
    zo, saved = O.fm_encoder_fwd(p, dd, 'cat')
    go = O.fm_encoder_bwd(up, p, dd, saved, 'cat')
    close_act(z, zo)
    for k, prm in enc.named_parameters():
        close_grad(prm.grad, go[k])


@pytest.fixture(scope='module')
def biosnap_full():
    from tip_amd.data import build_data_dict
    return build_data_dict()


def test_graph_handle_full_biosnap_both_layers_match_the_modules(biosnap_full):
    """BASELINE config 2's D-D graph at full size (645 drugs, 1 097 relations, 8.3 M edges) through the op-level handle: both
    R-GCN layers (64 -> 32 with the ReLU inside, 32 -> 16) forward + backward on plans the library built in C++, against the
    PyTorch modules on `plan.py`'s plans -- the same bits -- and the handle's GENERIC route (work-item gathers) within tolerance."""
    import ctypes as C
    from tip_amd import _lib
    from tip_amd.layers import MyRGCNConv2
    L = _lib.lib()
    dd = biosnap_full
    n, r = dd['n_drug'], dd['n_dd_et']
    ei, rg = dd['dd_train_idx'].to(DEV), dd['dd_train_range'].to(DEV)
    torch.manual_seed(3)
    m1, m2 = MyRGCNConv2(64, 32, r, 32, after_relu=False).to(DEV), MyRGCNConv2(32, 16, r, 32, after_relu=True).to(DEV)
    x = torch.randn(n, 64, device=DEV).requires_grad_(True)
    up = torch.randn(n, 16, device=DEV)
    h1 = torch.relu(m1(x, ei, None, rg))
    h1.retain_grad()
    z = m2(h1, ei, None, rg)
    z.backward(up)
    ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run(h, prepared):
        par = [[t.detach().contiguous() for t in (m.basis, m.att, m.root)] for m in (m1, m2)]
        ws = [torch.empty(L.tipk_rgcn_workspace_bytes(h, di, do, 32), dtype=torch.uint8, device=DEV) for di, do in ((64, 32), (32, 16))]
        o1, o2 = torch.empty(n, 32, device=DEV), torch.empty(n, 16, device=DEV)
        xd = x.detach()
        assert L.tipk_rgcn_fwd(h, ptr(xd), 64, 64, ptr(par[0][0]), ptr(par[0][1]), ptr(par[0][2]), 32, 32, 1, ptr(o1), 32, ptr(ws[0]),
                               ws[0].numel(), st) == 0
        assert L.tipk_rgcn_fwd(h, ptr(o1), 32, 32, ptr(par[1][0]), ptr(par[1][1]), ptr(par[1][2]), 32, 16, 0, ptr(o2), 16, ptr(ws[1]),
                               ws[1].numel(), st) == 0
        g2 = [torch.empty_like(t) for t in [o1] + par[1]]
        assert L.tipk_rgcn_bwd_ex(h, ptr(o1), 32, 32, ptr(par[1][0]), ptr(par[1][1]), ptr(par[1][2]), 32, 16, ptr(up), 16, None, 0,
                                  ptr(g2[0]), 32, ptr(g2[1]), ptr(g2[2]), ptr(g2[3]), ptr(ws[1]), ws[1].numel(), int(prepared), st) == 0
        g1 = [torch.empty_like(t) for t in [xd] + par[0]]
        assert L.tipk_rgcn_bwd_ex(h, ptr(xd), 64, 64, ptr(par[0][0]), ptr(par[0][1]), ptr(par[0][2]), 32, 32, ptr(g2[0]), 32, ptr(o1), 32,
                                  ptr(g1[0]), 64, ptr(g1[1]), ptr(g1[2]), ptr(g1[3]), ptr(ws[0]), ws[0].numel(), int(prepared), st) == 0
        return [o1, o2] + g2 + g1

    want = [h1.detach(), z.detach(), h1.grad, m2.basis.grad, m2.att.grad, m2.root.grad, x.grad, m1.basis.grad, m1.att.grad, m1.root.grad]
    h = C.c_void_p()
    assert L.tipk_graph_build(ptr(ei), None, ptr(rg), 8, ei.shape[1], n, r, None, C.byref(h)) == 0
    try:
        generic = run(h, False)
        for got, ref in zip(generic, want):
            close(got, ref, rtol=1e-4, atol=2e-5 * float(ref.abs().max()))
        assert L.tipk_graph_prepare_rgcn(h, 32, 32) == 0 and L.tipk_graph_prepare_rgcn(h, 32, 16) == 0
        assert L.tipk_graph_rgcn_route(h, 32, 32) == 2 and L.tipk_graph_rgcn_route(h, 32, 16) == 2
        for got, ref in zip(run(h, True), want):
            assert torch.equal(got, ref)
    finally:
        assert L.tipk_graph_destroy(h) == 0


@pytest.mark.parametrize('mod,generic', [('cat', False), ('add', False), ('cat', True)])
def test_full_biosnap_encoder_vs_oracle(biosnap_full, mod, generic, monkeypatch):
    """BASELINE configs 2 and 3 at full size: z and all parameter gradients vs the CPU oracle.
    `generic` forces the fabric-gather kernels (the path large graphs take) instead of the
    LDS-resident relation-local ones."""
    if generic:
        monkeypatch.setenv('TIPK_NO_RELLOCAL', '1')
    from tip_amd.data import Data
    from tip_amd.layers import FMEncoder
    dd = biosnap_full
    R = dd['n_dd_et']
    dims = dict(prot_drug_dim=16, n_embed=48) if mod == 'cat' else dict(prot_drug_dim=64, n_embed=64)
    p = O.init_params(dd['n_drug'], dd['n_prot'], R, mod=mod, seed=1111, **dims)
    enc = FMEncoder(DEV, dd['n_drug'], R, dd['n_prot'], dd['n_prot'], dd['n_drug'], num_base=32, n_hid1=32,
                    n_hid2=16, mod=mod, **dims)
    enc = load_params(enc, p)
    d = Data.from_dict(dd).to(DEV)
    z = enc(d.d_feat, d.dd_train_idx, d.dd_train_et, d.dd_train_range, d.d_norm, d.p_feat, d.pp_train_indices,
            d.dp_edge_index, d.dp_range_list)
    torch.manual_seed(0)
    up = torch.randn(dd['n_drug'], 16)
    (z * up.to(DEV)).sum().backward()
    zo, saved = O.fm_encoder_fwd(p, dd, mod)
    go = O.fm_encoder_bwd(up, p, dd, saved, mod)
    close_act(z, zo)
    for k, prm in enc.named_parameters():
        close_grad(prm.grad, go[k])


def test_paper_configuration_963_relations_encoder_and_objective_vs_oracle():
    """The paper's configuration (BASELINE config 2 "~960 relations": the 963 side effects with >= 500 drug pairs,
    reference analysis/evaluation.ipynb) at full size: z, every encoder gradient, the fused objective and
    d decoder.weight against the CPU oracle -- `bench.py --workload biosnap963` times exactly this graph."""
    from tip_amd.data import build_data_dict, Data
    from tip_amd.layers import FMEncoder, MultiInnerProductDecoder
    from tip_amd.neg_sampling import typed_negative_sampling
    dd = build_data_dict(min_pairs=500)
    R = dd['n_dd_et']
    assert R == 963
    dims = dict(prot_drug_dim=16, n_embed=48)
    p = O.init_params(dd['n_drug'], dd['n_prot'], R, mod='cat', seed=1111, **dims)
    enc = FMEncoder(DEV, dd['n_drug'], R, dd['n_prot'], dd['n_prot'], dd['n_drug'], num_base=32, n_hid1=32,
                    n_hid2=16, mod='cat', **dims)
    enc = load_params(enc, p)
    dec = MultiInnerProductDecoder(16, R)
    dec.weight.data = p['decoder.weight'].clone()
    dec = dec.to(DEV)
    d = Data.from_dict(dd).to(DEV)
    z = enc(d.d_feat, d.dd_train_idx, d.dd_train_et, d.dd_train_range, d.d_norm, d.p_feat, d.pp_train_indices,
            d.dp_edge_index, d.dp_range_list)
    neg = typed_negative_sampling(d.dd_train_idx, d.n_drug, d.dd_train_range, seed=11)
    loss = dec.objective(z, d.dd_train_idx, neg, d.dd_train_et)
    loss.backward()
    zo, saved = O.fm_encoder_fwd(p, dd, 'cat')
    negc = neg.cpu()
    ps = O.distmult_fwd(zo, dd['dd_train_idx'], dd['dd_train_et'], p['decoder.weight'])
    ns = O.distmult_fwd(zo, negc, dd['dd_train_et'], p['decoder.weight'])
    gp, gn = O.tip_loss_bwd(ps, ns)
    gz1, gw1 = O.distmult_bwd(gp, zo, dd['dd_train_idx'], dd['dd_train_et'], p['decoder.weight'])
    gz2, gw2 = O.distmult_bwd(gn, zo, negc, dd['dd_train_et'], p['decoder.weight'])
    go = O.fm_encoder_bwd(gz1 + gz2, p, dd, saved, 'cat')
    close_act(z, zo)
    close(loss, O.tip_loss(ps, ns), rtol=1e-4, atol=1e-6)
    close_grad(dec.weight.grad, gw1 + gw2, scale=10)        # (8.3 M fixed-point terms against the oracle's fp32 index_add: 6e-5 seen)
    for k, prm in enc.named_parameters():
        close_grad(prm.grad, go[k], scale=5)                     # (behind the decoder's gradient: 9e-6 seen)


def test_full_biosnap_size_independent_properties(biosnap_full):
    """Linearity and symmetry of the D-D aggregation at full size (no oracle needed):
    rgcn(a x1 + b x2) = a rgcn(x1) + b rgcn(x2);  <A x, y> = <x, A^T y> via autograd."""
    from tip_amd.layers import MyRGCNConv2
    dd = biosnap_full
    R, N = dd['n_dd_et'], dd['n_drug']
    torch.manual_seed(1)
    m = MyRGCNConv2(64, 32, R, 32, after_relu=False).to(DEV)
    ei, et, rg = dd['dd_train_idx'].to(DEV), dd['dd_train_et'].to(DEV), dd['dd_train_range'].to(DEV)
    x1, x2 = torch.randn(N, 64, device=DEV), torch.randn(N, 64, device=DEV)
    with torch.no_grad():
        lhs = m(0.3 * x1 - 1.7 * x2, ei, et, rg)
        rhs = 0.3 * m(x1, ei, et, rg) - 1.7 * m(x2, ei, et, rg)
    close(lhs, rhs, rtol=1e-4, atol=1e-5)
    x = x1.clone().requires_grad_(True)
    y = torch.randn(N, 32, device=DEV)
    out = m(x, ei, et, rg)
    (out * y).sum().backward()
    # adjoint identity: <J x', y> == <x', J^T y> for a fresh direction x'
    with torch.no_grad():
        jx = m(x2, ei, et, rg)
    a = float((jx.double() * y.double()).sum())
    b = float((x2.double() * x.grad.double()).sum())
    assert abs(a - b) <= 1e-4 * max(1.0, abs(a)), (a, b)


# ------------------------------------------------------------------ multi-rank (2 processes, one GPU, gloo)
def _shard_worker(rank, world, port, ret, max_relations=12):
    """One rank of the relation-sharded TRAINING STEP: local edges, local att /
    decoder.weight rows, sharded sampler + objective, flat-buffer collectives -- against the unsharded
    model in the same process: loss, every gradient (local rows vs the matching rows), parameters after
    an Adam step, embeddings and the gathered test() record."""
    import os
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from tip_amd.data import build_data_dict
        from tip_amd.dist import make_shard, shard_state_dict, gather_state_dict, shard_edges, LOCAL_ROWS
        from tip_amd.layers import TIP, Setting
        from tip_amd.neg_sampling import typed_negative_sampling
        import time
        t_start = time.perf_counter()

        def stamp(what):                                                   # TIPK_TEST_STAMPS=1 pytest -s: where a rank's time goes
            if os.environ.get('TIPK_TEST_STAMPS'):
                torch.cuda.synchronize()
                print('[rank %d] %7.1f s  %s' % (rank, time.perf_counter() - t_start, what), flush=True)
        dd = build_data_dict(max_relations=max_relations)
        R = dd['n_dd_et']
        st = Setting()
        from tip_amd import neg_sampling as NS
        stamp('data dict')
        # the UNSHARDED model runs on rank 0 only (8 ranks share one GPU: eight full-size references cost 9 minutes);
        # what the other ranks compare against travels through the process group
        ref_pack = [None]
        if rank == 0:
            torch.manual_seed(3)
            NS.manual_seed(77)
            ref = TIP(st, torch.device(DEV), data=dd)
            stamp('reference model built')
            full_sd = {k: v.detach().cpu().clone() for k, v in ref.state_dict().items()}
            neg = typed_negative_sampling(ref.data.dd_train_idx, ref.data.n_drug, ref.data.dd_train_range, seed=5)
            opt_r = torch.optim.Adam(ref.parameters(), lr=0.01)
            steps_r = []
            for step in range(2):
                opt_r.zero_grad()
                loss_r = ref(neg_index=neg)
                loss_r.backward()
                steps_r.append({'loss': float(loss_r), 'emb': ref.embeddings.detach().cpu().clone(),
                                'grads': {k: p.grad.detach().cpu().clone() for k, p in ref.named_parameters()}})
                opt_r.step()
                stamp('reference step %d' % step)
            ref_pack = [{'full_sd': full_sd, 'test_neg': ref.test_neg_index.cpu(), 'neg': neg.cpu(), 'steps': steps_r,
                         'final_sd': {k: v.detach().cpu().clone() for k, v in ref.state_dict().items()},
                         'rec': ref.test(print_output=False)}]
            stamp('reference test()')
            del ref, opt_r
            torch.cuda.empty_cache()
        dist.broadcast_object_list(ref_pack, src=0)
        stamp('reference results received')
        rp = ref_pack[0]
        full_sd, neg = rp['full_sd'], rp['neg'].to(DEV)
        ref_test_neg = rp['test_neg'].to(DEV)
        shard = make_shard(dd['dd_train_range'], rank, world)
        assert shard.rel_ids.numel() < R and (world > R or shard.rel_ids.numel() > 0)
        NS.manual_seed(77)
        model = TIP(st, torch.device(DEV), data=dd, shard=shard)           # this rank's relations only
        stamp('sharded model built')
        # the sampler's counters run over GLOBAL positions: a rank draws the unsharded run's negatives of its relations
        assert torch.equal(model.test_neg_index, shard_edges(ref_test_neg, dd['dd_test_range'], shard.rel_ids)[0])
        assert model.data.dd_train_idx.shape[1] == shard.n_train_local < shard.n_train_total
        assert model.decoder.weight.shape[0] == model.encoder.rgcn1.att.shape[0] == shard.rel_ids.numel()
        model.load_state_dict(shard_state_dict(full_sd, shard))
        model.test_neg_index, _ = shard_edges(ref_test_neg, dd['dd_test_range'], shard.rel_ids)
        neg_l, _ = shard_edges(neg, dd['dd_train_range'], shard.rel_ids)
        opt_s = torch.optim.Adam(model.parameters(), lr=0.01)
        ok = True
        for step in range(2):
            opt_s.zero_grad()
            want_step = rp['steps'][step]
            loss_s = model(neg_index=neg_l.contiguous())
            loss_s.backward()
            ok = ok and abs(want_step['loss'] - float(loss_s)) <= 1e-5 * abs(want_step['loss'])
            ok = ok and torch.allclose(model.embeddings.cpu(), want_step['emb'], rtol=1e-4, atol=1e-5)
            for k, p in model.named_parameters():
                want = want_step['grads'][k]
                if k in LOCAL_ROWS:
                    want = want[shard.rel_ids.cpu()]
                if want.numel() == 0:                                  # (a rank without relations: its att / decoder rows are empty)
                    good = p.grad is None or p.grad.numel() == 0
                else:
                    good = torch.allclose(p.grad.cpu(), want, rtol=1e-4, atol=2e-5 * max(1e-6, float(want.abs().max())))
                if not good:
                    print('rank', rank, 'step', step, 'grad mismatch', k, flush=True)
                ok = ok and good
            opt_s.step()
            stamp('sharded step %d' % step)
        # after two Adam steps: gathered full state == the unsharded model's state
        got = gather_state_dict(model, shard)
        for k, v in rp['final_sd'].items():
            good = torch.allclose(got[k], v, rtol=0, atol=2e-3)            # 20 % of one Adam step of 0.01
            if not good:
                print('rank', rank, 'state mismatch', k, float((got[k] - v).abs().max()), flush=True)
            ok = ok and good
        rec_r = rp['rec']
        stamp('state gathered')
        rec_s = model.test(print_output=False)
        stamp('sharded test()')
        ok = ok and rec_s.shape == (3, R) and bool(np.allclose(rec_s, rec_r, atol=2e-4))
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def _run_shard_workers(world, max_relations, env=None):
    import os
    import socket
    import torch.multiprocessing as mp
    saved = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})                                     # (spawned ranks inherit it: ops reads TIPK_FWD_ROUTE at import)
    try:
        _run_shard_workers_(world, max_relations)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _run_shard_workers_(world, max_relations):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, ret, max_relations)) for r in range(world)]
    from conftest import start_ranks
    start_ranks(procs)
    for p in procs:
        p.join(800 if world <= 4 else 1400)
        assert p.exitcode == 0
    assert dict(ret) == {r: True for r in range(world)}


@pytest.mark.timeout(600)
def test_sharded_training_step_two_ranks():
    _run_shard_workers(2, 12)


@pytest.mark.timeout(600)
def test_sharded_step_with_a_rank_that_holds_no_relation():
    """More ranks than relations (2 relations over 3 ranks: rank 2 holds none) with the TIMED forward-route decision: the rank
    without relations still takes part in every collective of the step -- the route-timing all-reduce included -- and the
    job's loss, embeddings, gradients, parameters after two Adam steps and test() record equal the unsharded model's."""
    _run_shard_workers(3, 2, env={'TIPK_FWD_ROUTE': 'timed'})


@pytest.mark.timeout(900)
def test_config4_biosnap_sharded_over_four_ranks():
    """BASELINE config 4 (TIP-cat, relations sharded by relation id) over 4 ranks sharing the one GPU of the test box over
    gloo, FULL size (all 1 097 relations): every rank's loss, embeddings, gradients (shard-local rows against the matching
    rows), the parameters after two Adam steps and the gathered test() record equal the unsharded model's."""
    _run_shard_workers(4, None)


@pytest.mark.timeout(1500)
def test_config4_full_biosnap_sharded_over_eight_ranks():
    """BASELINE config 4 at the rank count it names and at FULL size: all 1 097 relations over EIGHT ranks (sharing the one
    GPU of the test box over gloo; 8 x MI355X over RCCL is the driver's to run).  With 8 shards the partition, the per-rank
    plan sizes and the forward route (>= 4 ranks: Y route) differ from the 4-rank case; same assertions: loss, embeddings,
    gradients, two Adam steps, the gathered test() record."""
    _run_shard_workers(8, None)


def test_graphed_train_step_matches_eager():
    """tip_amd.train.GraphedTrainStep (whole step in one hipGraph, sampler counter on device) gives
    the same losses as the eager loop with the same seeds."""
    from tip_amd import neg_sampling as NS
    from tip_amd.data import build_data_dict
    from tip_amd.layers import TIP, Setting
    from tip_amd.train import GraphedTrainStep
    dd = build_data_dict(max_relations=6)
    st = Setting(sp_rate=0.9, lr=0.01, prot_drug_dim=16, n_embed=48, n_hid1=32, n_hid2=16, num_base=32)
    losses = []
    for graphed in (False, True):
        torch.manual_seed(5)
        model = TIP(st, torch.device(DEV), data=dd)
        from tip_amd.optim import Adam
        opt = Adam(model.parameters(), lr=st.lr)
        NS.manual_seed(123)
        out = []
        if graphed:
            step = GraphedTrainStep(model, opt, warmup=2)          # 2 warm-up steps + 1 captured (not run)
            for _ in range(3):
                out.append(float(step()))
        else:
            for i in range(5):
                opt.zero_grad(set_to_none=True)
                loss = model()
                loss.backward()
                opt.step()
                if i >= 2:
                    out.append(float(loss))
        losses.append(out)
    # the sampler's position and seed live on the device (capture executes nothing, so it consumes no
    # position): steps 3-5 draw the same negatives either way and the trajectories agree step by step
    assert all(1.0 < v < 1.45 for v in losses[0] + losses[1]), losses
    assert losses[0] == losses[1], losses                              # bit for bit: no float atomics anywhere in the step
    # two epochs per replayed graph: after ONE replay the model stands where the one-epoch graph stands after two
    torch.manual_seed(5)
    model = TIP(st, torch.device(DEV), data=dd)
    from tip_amd.optim import Adam
    opt = Adam(model.parameters(), lr=st.lr)
    NS.manual_seed(123)
    step2 = GraphedTrainStep(model, opt, warmup=2, steps_per_replay=2)
    assert float(step2()) == losses[1][1]                              # epochs 3 and 4 of the trajectory: the loss of the 4th


def test_whole_model_pickle_roundtrip_after_forward(tmp_path):
    """`torch.save(model, path)` as the reference's tip.py:36 works after training steps (plan caches
    hold device plans and closures: dropped on pickling, rebuilt on first use) and the reloaded model
    computes the same loss with the same negatives."""
    from tip_amd.data import build_data_dict
    from tip_amd.layers import TIP, Setting
    from tip_amd.neg_sampling import typed_negative_sampling
    dd = build_data_dict(max_relations=5)
    torch.manual_seed(3)
    model = TIP(Setting(), torch.device(DEV), data=dd)
    d = model.data
    neg = typed_negative_sampling(d.dd_train_idx, d.n_drug, d.dd_train_range, seed=5)
    loss = model(neg)
    loss.backward()
    rec = model.test(print_output=False)
    path = str(tmp_path / 'tip-cat-example.pt')
    torch.save(model, path)
    back = torch.load(path, weights_only=False)
    assert back.encoder.rgcn1._cache.value is None                   # caches were not pickled
    loss2 = back(neg)
    assert torch.equal(back.embeddings, model.embeddings)             # every kernel of the step is bitwise reproducible
    assert torch.equal(loss2.detach(), loss.detach())
    assert np.array_equal(back.test(print_output=False), rec)


def test_reference_pickle_drop_in():
    """north_star: "the reference's data_dict.pkl ... run unchanged".  tests/golden/data_dict_small.pkl was
    written by the reference's OWN prepare.py:10-47 pipeline (oracle/make_pickle_fixture.py executes it on a
    reduced data directory: 6 relations, 500 proteins; scipy matrices and per-relation lists kept, and the
    pipeline's quirk that isolated drugs are dropped from the adjacency but not from n_drug).  The HIP `TIP`
    loads it through the reference's constructor signature and must reproduce what the reference's `TIP`
    computed on the same file with the same weights and negatives: loss, every gradient, `test()` record
    (src/layers.py:284-293, :328-375)."""
    import os
    from conftest import GOLDEN
    from tip_amd.layers import TIP, Setting
    g = load_golden('tip_from_pickle')
    path = os.path.join(GOLDEN, 'data_dict_small.pkl')
    st = Setting(sp_rate=0.9, lr=0.01, prot_drug_dim=16, n_embed=48, n_hid1=32, n_hid2=16, num_base=32)   # tip.py:14
    model = TIP(st, torch.device(DEV), data_path=path)                 # tip.py:15
    assert model.data_source.endswith('data_dict_small.pkl')
    d = model.data
    assert (d.n_drug, d.n_prot, d.n_dd_et) == (int(g['n_drug']), int(g['n_prot']), int(g['n_dd_et']))
    assert d.dd_train_idx.shape[1] == int(g['n_train']) and d.dd_train_idx.is_cuda
    sd = model.state_dict()
    assert sorted(sd) == sorted(k[len('param.'):] for k in g if k.startswith('param.'))
    model.load_state_dict({k: g['param.' + k] for k in sd})
    model.test_neg_index = g['test_neg'].to(DEV)                       # the reference drew these with numpy's global RNG
    loss = model(neg_index=g['train_neg'].to(DEV))
    close(loss, g['loss'], rtol=2e-5, atol=1e-6)
    close_act(model.embeddings, g['embeddings'])
    loss.backward()
    for k, prm in model.named_parameters():
        want = g['grad.' + k]
        close_grad(prm.grad, want, scale=5)
    rec = model.test(print_output=False)
    np.testing.assert_allclose(rec, g['record'].numpy(), rtol=0, atol=2e-4)   # AUPRC/AUROC/AP per relation
    # the unfused loss path (decoder scores + torch ops, exactly src/layers.py:335-340) agrees too
    model.fused_loss = False
    close(model(neg_index=g['train_neg'].to(DEV)), g['loss'], rtol=2e-5, atol=1e-6)


def test_nn_decoder_golden():
    """NNDecoder against the reference's own forward and autograd gradients."""
    from tip_amd.layers import NNDecoder
    g = load_golden('nn_decoder')
    m = load_params(NNDecoder(6, g['w1_l2'].shape[0], l1_dim=5), g)
    z = g['z'].to(DEV).requires_grad_(True)
    s = m(z, g['dd_idx'].to(DEV), g['dd_et'].to(DEV))
    close(s, g['score'])
    (s * g['upstream'].to(DEV)).sum().backward()
    close(z.grad, g['grad_z'], atol=1e-5)
    for k in ('w1_l1', 'w1_l2', 'w2_l1', 'w2_l2'):
        close(getattr(m, k).grad, g['grad.' + k], atol=1e-5)


def test_rgcn_backward_after_another_forward_recomputes_xb():
    """The pair-form forward keeps XB in a buffer that belongs to the GRAPH (the next forward pass rewrites it); a backward
    pass that runs after another forward must not read the other pass's XB (ops._RGCN: stamp -> recompute)."""
    from tip_amd.layers import MyRGCNConv2
    g = torch.Generator().manual_seed(3)
    N, R, d_in, d_out = 40, 6, 8, 16
    half = [torch.randint(0, N, (2, 30), generator=g) for _ in range(R)]
    ei = torch.cat([torch.cat([h, h.flip(0)], 1) for h in half], 1).to(DEV)
    sizes = torch.tensor([60] * R)
    end = torch.cumsum(sizes, 0)
    rg = torch.stack([end - sizes, end], 1)
    et = torch.repeat_interleave(torch.arange(R), sizes).to(DEV)
    layer = MyRGCNConv2(d_in, d_out, R, 32, after_relu=False).to(DEV)
    x1 = torch.randn(N, d_in, generator=g).to(DEV).requires_grad_()
    x2 = (torch.randn(N, d_in, generator=g) * 3).to(DEV).requires_grad_()
    gup = torch.randn(N, d_out, generator=g).to(DEV)

    def grads(x, disturb):
        layer.zero_grad()
        x.grad = None
        out = layer(x, ei, et, rg)
        if disturb == 'grad':
            layer(x2, ei, et, rg)                                         # rewrites the graph's cell / XB buffers
        elif disturb == 'no_grad':                                        # ... as does a pass that saves nothing:
            with torch.no_grad():                                         # other x AND other att -> other cells
                keep = layer.att.data.clone()
                layer.att.data.mul_(-2.0)
                layer(x2, ei, et, rg)
                layer.att.data.copy_(keep)
        out.backward(gup)
        return [x.grad.clone()] + [p.grad.clone() for p in layer.parameters()]
    clean = grads(x1, None)
    for mode in ('grad', 'no_grad'):
        for a, b in zip(clean, grads(x1, mode)):
            torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-6)


@pytest.mark.timeout(900)
def test_nn_decoder_objective_biosnap_size():
    """NNDecoder as a TRAINING decoder (src/layers.py:598-637 as the objective of model/ddm-nn.py:65-102) at
    BioSNAP size: the fused objective on the transposed tables (`tipk_pair_table_loss`) == the oracle's literal forward +
    explicit backward over all 8.3 M positive and 8.3 M negative triples (loss and all five gradients), == the unfused path
    (scores + torch ops), and is reproducible bit for bit."""
    from tip_amd.data import build_data_dict
    from tip_amd.layers import NNDecoder
    from tip_amd.neg_sampling import typed_negative_sampling
    dd = build_data_dict()
    n, R = dd['n_drug'], dd['n_dd_et']
    g = torch.Generator().manual_seed(17)
    z_c = torch.randn(n, 16, generator=g) * 0.5
    m = NNDecoder(16, R, l1_dim=16)
    for prm in m.parameters():
        prm.data = torch.randn(prm.shape, generator=g) * 0.3
    w = {k: v.detach().clone() for k, v in m.named_parameters()}
    m = m.to(DEV)
    pos, et, rg = dd['dd_train_idx'].to(DEV), dd['dd_train_et'].to(DEV), dd['dd_train_range'].to(DEV)
    neg = typed_negative_sampling(pos, n, rg, seed=5)

    def run():
        m.zero_grad()
        z = z_c.to(DEV).requires_grad_(True)
        loss = m.objective(z, pos, neg, et)
        loss.backward()
        return loss.detach(), [z.grad.clone()] + [getattr(m, k).grad.clone() for k in ('w1_l1', 'w1_l2', 'w2_l1', 'w2_l2')]
    loss, grads = run()
    loss2, grads2 = run()
    assert torch.equal(loss, loss2) and all(torch.equal(a, b) for a, b in zip(grads, grads2))
    # oracle: literal forward, the loss of src/layers.py:335-340, explicit backward
    posc, negc, etc = dd['dd_train_idx'], neg.cpu(), dd['dd_train_et']
    args = (w['w1_l1'], w['w1_l2'], w['w2_l1'], w['w2_l2'])
    ps, ns = O.nn_decoder_fwd(z_c, posc, etc, *args), O.nn_decoder_fwd(z_c, negc, etc, *args)
    close(loss, O.tip_loss(ps, ns).view(1), rtol=2e-5, atol=1e-6)
    gp, gn = O.tip_loss_bwd(ps, ns)
    want = [a + b for a, b in zip(O.nn_decoder_bwd(gp, z_c, posc, etc, *args), O.nn_decoder_bwd(gn, z_c, negc, etc, *args))]
    for got, ref in zip(grads, want):
        close_grad(got, ref, scale=5)                           # (the NNDecoder's score tables at BioSNAP size: 9e-6 seen)
    # the unfused path of the module (forward() scores + torch ops) agrees
    z = z_c.to(DEV)
    unf = -torch.log(m(z, pos, et) + 1e-13).mean() - torch.log(1 - m(z, neg, et) + 1e-13).mean()
    close(unf.view(1), loss, rtol=2e-5, atol=1e-6)
