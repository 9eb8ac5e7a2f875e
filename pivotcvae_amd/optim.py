"""Flat-buffer Adam (K8) - the optimiser of reference train_generative.py:103 (torch.optim.Adam, defaults).

All trainable parameters are re-homed into ONE contiguous fp32 buffer (their nn.Parameter objects, names and
state_dict keys are unchanged: each becomes a view), and so are their gradients.  One step is then one
memset + one HIP kernel, and data parallelism needs exactly one all-reduce over the gradient buffer.

Parameters that never receive a gradient (the PSM stack) keep a zero gradient; Adam with
g = 0, m = 0, v = 0 leaves them bit-identical, which is what the reference's "skip grad None" does.  With a
non-zero weight_decay that is no longer automatic (g + wd * p != 0): torch.optim.Adam skips a parameter whose
grad is None altogether, so the ranges of such parameters (``model.params_without_grad()``) are stepped with
weight_decay = 0 - the flat buffer is then walked as a few contiguous segments instead of one.
"""
import torch

from . import ops


class FlatAdam:
    TAIL = 4

    def __init__(self, model, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        # weight_decay: torch.optim.Adam's L2 term (pretrain_env.py:59); train_generative.py:103 passes none
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), betas, float(eps), float(weight_decay)
        params = [p for p in model.parameters() if p.requires_grad]
        if not params:
            raise ValueError("no trainable parameters")
        # parameters the model wants ADJACENT in the flat buffer (the two heads of a stack: their weights back to back are one
        # [2 Z, K] operand, so mu and logvar come out of ONE GEMM - models' flat_param_groups()); everything else in module order
        groups = [[q for q in g if q.requires_grad] for g in getattr(model, "flat_param_groups", lambda: [])()]
        first = {id(g[0]): g for g in groups if g}
        placed, self.params = set(), []
        for p in params:
            for q in first.get(id(p), [p]):
                if id(q) not in placed:
                    placed.add(id(q))
                    self.params.append(q)
        dev = self.params[0].device
        total = sum(p.numel() for p in self.params)
        self.flat = torch.empty(total, dtype=torch.float32, device=dev)
        # the gradient buffer carries a short tail behind the parameters' gradients: a data-parallel trainer puts the step's
        # logged scalars there, so that ONE all-reduce per step moves gradients and statistics (Adam never sees the tail)
        self.grad_ext = torch.zeros(total + self.TAIL, dtype=torch.float32, device=dev)
        self.grad = self.grad_ext[:total]
        self.tail = self.grad_ext[total:]
        self.m = torch.zeros(total, dtype=torch.float32, device=dev)
        self.v = torch.zeros(total, dtype=torch.float32, device=dev)
        no_grad = {id(p) for p in getattr(model, "params_without_grad", lambda: [])()}
        off = 0
        self.segments = []   # [offset, numel, weight_decay], adjacent ranges of equal decay merged
        for p in self.params:
            n = p.numel()
            self.flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + n].view(p.shape)
            p.grad = self.grad[off:off + n].view(p.shape)
            wd = 0.0 if id(p) in no_grad else self.weight_decay
            if self.segments and self.segments[-1][2] == wd:
                self.segments[-1][1] += n
            else:
                self.segments.append([off, n, wd])
            off += n
        self.t = 0

    def zero_grad(self):
        if self.grad_ext.is_cuda:
            ops.zero_(self.grad_ext)   # one fill launch on the stream (gradients + the statistics tail); NOT a memset: csrc/elementwise.hip
        else:
            self.grad_ext.zero_()      # the CPU tests' optimiser (collective logic only)
        for p, g in zip(self.params, self._grad_views()):
            if p.grad is None or p.grad.data_ptr() != g.data_ptr():
                p.grad = g  # someone set it to None / replaced it: re-attach the flat view

    def _grad_views(self):
        off = 0
        for p in self.params:
            n = p.numel()
            yield self.grad[off:off + n].view(p.shape)
            off += n

    def step(self, grad_scale=1.0):
        self.t += 1
        for off, n, wd in self.segments:   # one launch unless weight decay is on AND some parameters never get a gradient
            sl = slice(off, off + n)
            ops.adam_step_(self.flat[sl], self.grad[sl], self.m[sl], self.v[sl], self.lr, self.t, self.betas[0], self.betas[1],
                           self.eps, grad_scale, wd)
