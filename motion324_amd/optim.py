"""Optimizer step and gradient plumbing of the reference's training loop on libm324 kernels.

Reference semantics reproduced (train.py:171-219, utils/training_utils.py:38-82):
  * AdamW(lr, betas=(beta1, beta2), eps 1e-8, weight_decay on parameters with dim() > 1 only, fused);
  * gradients: nan_to_num(nan=0, posinf=1e-6, neginf=-1e-6), global L2 norm, clip to grad_clip_norm,
    skip the update when the pre-clip norm exceeds allowed_gradnorm_factor * grad_clip_norm or the loss is not finite;
  * cosine schedule with linear warm-up (transformers.get_cosine_schedule_with_warmup).
Data parallelism: gradients live in ONE flat fp32 buffer (628 MB for the full model) that is all-reduced with a
single RCCL call per step (xGMI is point-to-point: few large collectives), then averaged.
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, List, Optional

import torch
import torch.distributed as dist

from . import ops
from .backward import GradStore
from .prepared import bump_generation


def cosine_with_warmup(step: int, warmup: int, total: int, base_lr: float) -> float:
    """lr at optimizer step `step` (0-based), as transformers.get_cosine_schedule_with_warmup(num_cycles=0.5)."""
    if step < warmup:
        return base_lr * step / max(1, warmup)
    prog = (step - warmup) / max(1, total - warmup)
    return base_lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * prog)))


class FusedAdamW:
    def __init__(self, named_params, lr=4e-4, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.05, grad_clip_norm=1.0,
                 allowed_gradnorm_factor=5.0, group=None):
        self.params: List[torch.nn.Parameter] = [p for _, p in named_params if p.requires_grad]
        self.decay = [p.dim() > 1 for p in self.params]                # training_utils.py:39-47
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.clip, self.skip_factor, self.group = grad_clip_norm, allowed_gradnorm_factor, group
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        self.flat_grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        self.offsets, o = [], 0
        for p in self.params:
            self.offsets.append(o)
            o += p.numel()
        self.step_count = 0
        self._partial = torch.empty(1024, dtype=torch.float32, device=dev)
        self._sumsq = torch.zeros((), dtype=torch.float32, device=dev)
        self._gscale = torch.ones((), dtype=torch.float32, device=dev)

    def grad_view(self, i: int) -> torch.Tensor:
        p = self.params[i]
        return self.flat_grad[self.offsets[i]:self.offsets[i] + p.numel()]

    def load_grads(self, G: GradStore, accumulate: bool = False) -> None:
        """Copies a GradStore into the flat buffer (torch copies: data movement only)."""
        for i, p in enumerate(self.params):
            g = G.get(p)
            if g is None:
                if not accumulate:
                    self.grad_view(i).zero_()
                continue
            if accumulate:
                self.grad_view(i).add_(g.reshape(-1))
            else:
                self.grad_view(i).copy_(g.reshape(-1))

    def all_reduce_mean(self) -> None:
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=self.group)
            self.flat_grad.mul_(1.0 / dist.get_world_size(self.group))

    def step(self, lr: Optional[float] = None, loss_is_finite: bool = True) -> Dict[str, float]:
        """One optimizer update from self.flat_grad.  Returns {'grad_norm', 'skipped'} (one host sync, as the
        reference's clip_grad_norm_(...).item())."""
        ops.grad_sumsq(self.flat_grad, self._sumsq, self._partial, sanitize=True, accumulate=False)
        norm = float(self._sumsq.sqrt())
        skipped = (not loss_is_finite) or (not math.isfinite(norm)) or norm > self.skip_factor * self.clip
        if not skipped:
            self.step_count += 1
            self._gscale.fill_(min(1.0, self.clip / (norm + 1e-6)))
            lr = self.lr if lr is None else lr
            for i, p in enumerate(self.params):
                o, n = self.offsets[i], p.numel()
                ops.adamw_step(p.data.view(-1), self.flat_grad[o:o + n], self.m[o:o + n], self.v[o:o + n], lr, self.betas[0],
                               self.betas[1], self.eps, self.wd if self.decay[i] else 0.0, self.step_count, self._gscale)
            bump_generation()           # the kernel wrote parameter memory behind torch's back: drop cached bf16 copies
        return {"grad_norm": norm, "skipped": skipped}
