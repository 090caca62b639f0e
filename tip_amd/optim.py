"""Adam for the training loop of tip.py:24-30 (`torch.optim.Adam(model.parameters(), lr=0.01)`): ONE launch per step.

`torch.optim.Adam(..., capturable=True, fused=True)` gives every workgroup 65536 elements of the parameter list: TIP-cat's
13 tensors (1.4 M floats) become 22 workgroups and 44 us + a 5 us step-counter launch of the 0.9 ms graphed epoch.
`tipk_adam_step` (include/tipk.h section 9) covers the list with 1024-element workgroups in one launch and counts its
steps in device memory, so a captured step (tip_amd/train.py) keeps counting on replay.

Same hyper-parameters, defaults, state names (`step`, `exp_avg`, `exp_avg_sq`) and update rule as torch.optim.Adam
(amsgrad / maximize / per-parameter lr tensors are not part of the reference's use and raise).
"""
import ctypes as C

import torch

from ._lib import check, lib, require_device, stream_ptr


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, maximize=False,
                 capturable=True):
        if amsgrad or maximize:
            raise ValueError('tip_amd.optim.Adam: amsgrad / maximize are not implemented (the reference uses neither)')
        if torch.is_tensor(lr):
            raise ValueError('tip_amd.optim.Adam: lr must be a number')
        if not 0.0 <= lr or not 0.0 <= eps or not 0.0 <= weight_decay:
            raise ValueError('invalid lr / eps / weight_decay')
        if not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError('invalid betas')
        # `capturable` is always true here (the step count lives on the device); the key is kept so that
        # tip_amd.train.GraphedTrainStep's check reads the same for both optimizers
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, capturable=True))
        self._tickets = {}                             # device -> int64 [528] (33 ticket words, a cache line each), 0 between launches
        self._counter_pool = {}                        # device -> (int64 [256] block, words handed out)

    def _new_counter(self, device):
        """A zeroed 0-dim int64 device tensor (views of one block per 256 parameters: no allocation per parameter)."""
        block, used = self._counter_pool.get(device, (None, 256))
        if used == 256:
            block, used = torch.zeros(256, dtype=torch.int64, device=device), 0
        self._counter_pool[device] = (block, used + 1)
        return block[used]

    def _counter_from(self, value, device):
        c = self._new_counter(device)
        c.fill_(int(value))
        return c

    def _adopt_state(self):
        """Step counts -> 0-dim int64 counters on the parameter's device.  torch's `Optimizer.load_state_dict` casts a
        capturable optimizer's `step` to a float32 tensor (and a `torch.optim.Adam` checkpoint holds a float tensor or a
        python number): `tipk_adam_step` reads the word as uint64, so such a value has to be re-made, not reinterpreted."""
        for p, st in self.state.items():
            if not st:
                continue
            stp = st.get('step')
            if stp is None:
                raise ValueError('tip_amd.optim.Adam: optimizer state without a step count')
            ok = torch.is_tensor(stp) and stp.dtype == torch.int64 and stp.dim() == 0 and stp.device == p.device
            if not ok:
                st['step'] = self._counter_from(round(float(stp)), p.device)
            for k in ('exp_avg', 'exp_avg_sq'):
                m = st[k]
                if m.dtype != torch.float32 or m.device != p.device or m.stride() != p.stride():
                    st[k] = torch.empty_like(p, memory_format=torch.preserve_format).copy_(m)

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        for g in self.param_groups:
            g['capturable'] = True
        self._adopt_state()

    def __setstate__(self, state):
        super().__setstate__(state)
        self.__dict__.setdefault('_tickets', {})
        self.__dict__.setdefault('_counter_pool', {})
        self._adopt_state()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            ps, gs, ms, vs = [], [], [], []
            for p in group['params']:
                if p.grad is None:
                    continue
                if p.dtype != torch.float32 or p.grad.is_sparse:
                    raise TypeError('tip_amd.optim.Adam: dense fp32 parameters only')
                require_device(p, p.grad)
                st = self.state[p]
                if not st:
                    st['step'] = self._new_counter(p.device)            # 0-dim int64 on the device, as with capturable=True
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                stp = st['step']
                if not (torch.is_tensor(stp) and stp.dtype == torch.int64 and stp.device == p.device):
                    raise TypeError('tip_amd.optim.Adam: state["step"] must be an int64 counter on the parameter\'s device '
                                    '(got %s): load checkpoints through load_state_dict()' % (
                                        '%s on %s' % (stp.dtype, stp.device) if torch.is_tensor(stp) else type(stp).__name__))
                g = p.grad
                if g.stride() != p.stride():                       # same dense layout as the parameter (and its moments)
                    g = torch.empty_like(p, memory_format=torch.preserve_format).copy_(g)
                if not _dense(p):
                    raise TypeError('tip_amd.optim.Adam: parameters must be dense (contiguous or a permutation of it)')
                ps.append(p); gs.append(g); ms.append(st['exp_avg']); vs.append(st['exp_avg_sq'])
            if not ps:
                continue
            dev = ps[0].device
            ticket = self._tickets.get(dev)
            if ticket is None:
                ticket = self._tickets[dev] = torch.zeros(528, dtype=torch.int64, device=dev)
            n = len(ps)
            arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
            numel = (C.c_int64 * n)(*[p.numel() for p in ps])
            b1, b2 = group['betas']
            check(lib().tipk_adam_step(n, arr(ps), arr(gs), arr(ms), arr(vs), numel, arr([self.state[p]['step'] for p in ps]),
                                       C.c_void_p(ticket.data_ptr()), float(group['lr']), float(b1), float(b2),
                                       float(group['eps']), float(group['weight_decay']), stream_ptr(dev)), 'tipk_adam_step')
        return loss


def _dense(t):
    """True if t's elements occupy numel() consecutive words (any permutation of a contiguous tensor)."""
    if t.is_contiguous():
        return True
    sizes_strides = sorted(((st, sz) for sz, st in zip(t.shape, t.stride()) if sz > 1))
    expect = 1
    for st, sz in sizes_strides:
        if st != expect:
            return False
        expect *= sz
    return True
