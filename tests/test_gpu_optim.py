"""tip_amd.optim.Adam (tipk_adam_step, include/tipk.h section 9) against torch.optim.Adam -- the optimizer of the
reference's training loop (tip.py:24-30)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _params(shapes, seed):
    g = torch.Generator().manual_seed(seed)
    out = []
    for s in shapes:
        if isinstance(s, tuple) and len(s) == 3 and s[2] == 't':      # a transposed (dense, non-contiguous) parameter
            out.append(torch.randn(s[0], s[1], generator=g).t())
        else:
            out.append(torch.randn(*((s,) if isinstance(s, int) else s), generator=g))
    return out


def _pair(shapes, seed, **kw):
    from tip_amd.optim import Adam
    init = _params(shapes, seed)
    # (clone of a transposed tensor keeps its strides: preserve_format)
    a = [torch.nn.Parameter(t.to(DEV).clone(memory_format=torch.preserve_format)) for t in init]
    b = [torch.nn.Parameter(t.to(DEV).clone(memory_format=torch.preserve_format)) for t in init]
    return a, b, Adam(a, **kw), torch.optim.Adam(b, foreach=False, fused=False, **kw)


@pytest.mark.parametrize('kw', [dict(lr=0.01), dict(lr=0.003, betas=(0.8, 0.99), eps=1e-6, weight_decay=0.01)])
def test_adam_matches_torch(kw):
    shapes = [1, 3, 1023, 1024, 1025, (645, 16), (32, 963, 't'), (963, 16), 70001, (7, 5, 3)]
    a, b, mine, ref = _pair(shapes, 5, **kw)
    assert not a[6].is_contiguous()
    g = torch.Generator().manual_seed(9)
    for it in range(7):
        for pa, pb in zip(a, b):
            gr = torch.randn(pa.shape, generator=g).to(DEV) * (0.1 + it)
            if it == 3 and pa.dim() == 2:
                gr = gr.t().contiguous().t()                          # a gradient whose layout differs from the parameter's
            pa.grad, pb.grad = gr.clone(memory_format=torch.preserve_format), gr.clone(memory_format=torch.preserve_format)
        if it == 5:
            a[0].grad = None; b[0].grad = None                        # a parameter without gradient is skipped
        mine.step(); ref.step()
        for pa, pb in zip(a, b):
            torch.testing.assert_close(pa, pb, rtol=1e-5, atol=2e-6)    # parameters ~ N(0, 1): a few ulps
    for pa, pb in zip(a[1:], b[1:]):
        torch.testing.assert_close(mine.state[pa]['exp_avg'], ref.state[pb]['exp_avg'], rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(mine.state[pa]['exp_avg_sq'], ref.state[pb]['exp_avg_sq'], rtol=1e-4, atol=1e-8)
        assert int(mine.state[pa]['step']) == 7 == int(ref.state[pb]['step'])
    assert set(mine.state_dict()['state'][1].keys()) == {'step', 'exp_avg', 'exp_avg_sq'}


def test_adam_long_list_takes_several_launches_and_counts_once():
    shapes = [17 + i for i in range(60)] + [5000]                     # 61 tensors: 48 + 13
    a, b, mine, ref = _pair(shapes, 2, lr=0.02)
    g = torch.Generator().manual_seed(1)
    for it in range(3):
        for pa, pb in zip(a, b):
            gr = torch.randn(pa.shape, generator=g).to(DEV)
            pa.grad, pb.grad = gr, gr.clone()
        mine.step(); ref.step()
    for pa, pb in zip(a, b):
        torch.testing.assert_close(pa, pb, rtol=1e-5, atol=2e-6)    # parameters ~ N(0, 1): a few ulps
    assert int(mine.state[a[0]]['step']) == 3 and int(mine.state[a[-1]]['step']) == 3


def test_adam_counts_steps_inside_a_captured_graph():
    """The step count lives on the device and is advanced by the launch itself: replays of a captured step keep counting
    (bias corrections change from replay to replay) -- same parameters as torch's eager Adam fed the same gradients."""
    shapes = [(645, 16), 1000, (963, 16)]
    a, b, mine, ref = _pair(shapes, 3, lr=0.01)
    grads = [torch.randn(p.shape, device=DEV) for p in a]
    for pa, pb, gr in zip(a, b, grads):
        pa.grad, pb.grad = gr, gr.clone()
    mine.step(); ref.step()                                           # warm-up step (allocates the state) outside the graph
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            mine.step()
    torch.cuda.synchronize()
    assert int(mine.state[a[0]]['step']) == 1                         # capture executes nothing
    for it in range(5):
        for gr in grads:
            gr.mul_(0.7).add_(0.05)                                   # new gradients, same storage
        for pb, gr in zip(b, grads):
            pb.grad = gr.clone()
        graph.replay(); ref.step()
    torch.cuda.synchronize()
    assert int(mine.state[a[0]]['step']) == 6
    for pa, pb in zip(a, b):
        torch.testing.assert_close(pa, pb, rtol=1e-5, atol=2e-6)    # parameters ~ N(0, 1): a few ulps


def test_adam_rejects_what_it_does_not_implement():
    from tip_amd.optim import Adam
    p = [torch.nn.Parameter(torch.zeros(4, device=DEV))]
    with pytest.raises(ValueError):
        Adam(p, amsgrad=True)
    with pytest.raises(ValueError):
        Adam(p, betas=(1.0, 0.9))
    cpu = [torch.nn.Parameter(torch.zeros(4))]
    cpu[0].grad = torch.ones(4)
    from tip_amd._lib import TipkError
    with pytest.raises(TipkError):
        Adam(cpu).step()                                              # no CPU path


def test_adam_state_dict_round_trip_and_torch_checkpoint():
    """`Optimizer.load_state_dict` casts a capturable optimizer's `step` to float32; the kernel reads the word
    as uint64.  A reloaded state (own checkpoint, or one written by torch.optim.Adam) must continue exactly like torch's
    Adam continues from the same checkpoint."""
    import copy
    from tip_amd.optim import Adam
    shapes = [(645, 16), 1000, (32, 963, 't'), 3]
    a, b, mine, ref = _pair(shapes, 11, lr=0.01)
    g = torch.Generator().manual_seed(4)

    def feed(params_a, params_b):
        for pa, pb in zip(params_a, params_b):
            gr = torch.randn(pa.shape, generator=g).to(DEV)
            pa.grad, pb.grad = gr.clone(memory_format=torch.preserve_format), gr.clone(memory_format=torch.preserve_format)
    for _ in range(7):
        feed(a, b)
        mine.step(); ref.step()
    sd_mine, sd_ref = copy.deepcopy(mine.state_dict()), copy.deepcopy(ref.state_dict())
    # three continuations: own checkpoint -> own optimizer, torch checkpoint -> own optimizer, torch -> torch (the reference)
    a1 = [torch.nn.Parameter(p.detach().clone(memory_format=torch.preserve_format)) for p in a]
    a2 = [torch.nn.Parameter(p.detach().clone(memory_format=torch.preserve_format)) for p in b]
    b2 = [torch.nn.Parameter(p.detach().clone(memory_format=torch.preserve_format)) for p in b]
    m1, m2 = Adam(a1, lr=0.01), Adam(a2, lr=0.01)
    r2 = torch.optim.Adam(b2, lr=0.01, foreach=False, fused=False)
    # (load_state_dict does not copy tensors that already have the right dtype and device: one deep copy per optimizer)
    m1.load_state_dict(sd_mine); m2.load_state_dict(copy.deepcopy(sd_ref)); r2.load_state_dict(copy.deepcopy(sd_ref))
    for opt, ps in ((m1, a1), (m2, a2)):
        for p in ps:
            st = opt.state[p]['step']
            assert st.dtype == torch.int64 and st.device == p.device and int(st) == 7
    for _ in range(3):
        for pa1, pa2, pb2 in zip(a1, a2, b2):
            gr = torch.randn(pa1.shape, generator=g).to(DEV)
            pa1.grad = gr.clone(memory_format=torch.preserve_format)
            pa2.grad = gr.clone(memory_format=torch.preserve_format)
            pb2.grad = gr.clone(memory_format=torch.preserve_format)
        m1.step(); m2.step(); r2.step()
    for pa1, pa2, pb2 in zip(a1, a2, b2):
        torch.testing.assert_close(pa2, pb2, rtol=1e-5, atol=2e-6)
        torch.testing.assert_close(pa1, pb2, rtol=1e-4, atol=2e-5)      # (a and b differed by a few ulps after 7 steps)
        assert int(m1.state[pa1]['step']) == 10 == int(m2.state[pa2]['step'])
    # a step count of the wrong type is refused, not reinterpreted
    m1.state[a1[0]]['step'] = torch.tensor(10.0, device=DEV)
    a1[0].grad = torch.zeros_like(a1[0])
    with pytest.raises(TypeError):
        m1.step()
