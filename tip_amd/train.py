"""Whole-step hipGraph for TIP training.

The reference's loop (`tip.py:24-30`) issues a few hundred small launches per epoch from Python; on
MI355X the kernels of one BioSNAP epoch take ~2 ms, the eager launches ~7 ms.  `GraphedTrainStep`
captures `zero_grad -> model() -> backward -> optimizer.step()` once (torch.cuda.graph = hipGraph) and
replays it: the negative sampler reads its stream position from device memory and advances it inside
the graph, so every replay draws new negatives (tip_amd/neg_sampling.py).

    opt = torch.optim.Adam(model.parameters(), lr=0.01, capturable=True)
    step = GraphedTrainStep(model, opt)
    for e in range(100):
        loss = step()            # 0-dim device tensor, valid until the next call
"""
import torch

from . import ops


class GraphedTrainStep(object):
    def __init__(self, model, optimizer, warmup=2, steps_per_replay=1):
        """steps_per_replay (> 1): that many consecutive epochs per replayed graph -- the hand-over between two replays (~9 us on
        MI355X) is paid once per replay; `__call__` then runs steps_per_replay epochs and returns the last one's loss."""
        for group in optimizer.param_groups:
            if not group.get('capturable', False):
                raise ValueError('construct the optimizer with capturable=True (its step counter must live on the device)')
        self.model, self.optimizer = model, optimizer
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                       # warm-up: builds every plan, allocator state
            for _ in range(warmup):
                self._step()
        torch.cuda.current_stream().wait_stream(side)
        # nothing may keep the warm-up autograd graph alive: its AccumulateGrad nodes are bound to the
        # warm-up stream and would be reused (and synchronised with) inside the capture
        if hasattr(model, 'embeddings') and torch.is_tensor(model.embeddings):
            model.embeddings = model.embeddings.detach()
        self.steps_per_replay = int(steps_per_replay)
        assert self.steps_per_replay >= 1
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            for _ in range(self.steps_per_replay):
                self.loss = self._step()

    def _step(self):
        self.optimizer.zero_grad(set_to_none=True)
        loss = self.model()
        # (a registered unit gradient: autograd's ones_like fill and the objective's scaling launch disappear)
        seed = ops.unit_grad(loss.device) if loss.dim() == 0 and loss.dtype == torch.float32 and loss.is_cuda else None
        loss.backward(gradient=seed)
        self.optimizer.step()
        return loss.detach()

    def __call__(self):
        self.graph.replay()
        return self.loss
