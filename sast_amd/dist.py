"""Data-parallel plumbing: flat parameter / gradient buffers, one gradient all-reduce per step over
RCCL (torch.distributed backend "nccl" on ROCm; "gloo" in CPU tests), fused AdamW on the flat buffers.

The hot path shards over the batch only: every rank runs the full model on its own
samples; the only exchange is the gradient all-reduce (the reference uses DDP, train.py:96-98).
BatchNorm statistics in the PAFPN stay rank-local (documented deviation from the reference's
SyncBatchNorm under DDP, train.py:167; identical to the reference's single-GPU numerics).
"""
from __future__ import annotations

from typing import Iterable, List

import torch
import torch.distributed as dist


def _phys_view(flat: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
    """view `flat` (1-D, numel == like.numel()) with the logical shape AND memory layout of `like`."""
    if like.dim() == 4 and not like.is_contiguous() and like.permute(0, 2, 3, 1).is_contiguous():
        co, ci, kh, kw = like.shape
        return flat.view(co, kh, kw, ci).permute(0, 3, 1, 2)
    if not like.is_contiguous():
        raise RuntimeError("unsupported parameter layout")
    return flat.view(like.shape)


class FlatParams:
    """re-homes every (unique) parameter of `modules` into one flat fp32 buffer and gives each a
    permanent `.grad` view into a second flat buffer.  The backward kernels accumulate straight into
    those views (functional._gbuf), so the whole gradient is ONE contiguous all-reduce message
    (75 MB for SAST: per-link cost on xGMI is paid once, not per bucket)."""

    def __init__(self, modules: Iterable[torch.nn.Module]):
        seen, params = set(), []
        for m in modules:
            for p in m.parameters():
                if id(p) not in seen and p.requires_grad:
                    seen.add(id(p))
                    params.append(p)
        self.params: List[torch.nn.Parameter] = params
        n = sum((p.numel() + 3) // 4 * 4 for p in params)
        dev = params[0].device
        self.flat = torch.zeros(n, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(n, device=dev, dtype=torch.float32)
        off = 0
        with torch.no_grad():
            for p in params:
                k = p.numel()
                v = _phys_view(self.flat[off:off + k], p)
                v.copy_(p)
                p.data = v
                p.grad = _phys_view(self.grad[off:off + k], p)
                off += (k + 3) // 4 * 4
        self.numel = n

    def zero_grad(self):
        """the ONLY supported way to clear gradients: `module.zero_grad()` / `optimizer.zero_grad()` (set_to_none=True) or
        `p.grad = None` would cut the views, the backward kernels would then accumulate into fresh tensors and the
        all-reduce / optimizer would keep reading a stale flat buffer (check_views catches that)."""
        self.grad.zero_()

    def check_views(self):
        """every parameter's .grad must still be its view into the flat gradient buffer"""
        base, end = self.grad.data_ptr(), self.grad.data_ptr() + 4 * self.numel
        for p in self.params:
            g = p.grad
            if g is None or not (base <= g.data_ptr() < end):
                raise RuntimeError("sast_amd.FlatParams: a parameter's .grad no longer points into the flat gradient buffer "
                                   "(zero_grad(set_to_none=True) or `p.grad = None`?): use FlatParams.zero_grad() only")

    def all_reduce(self, group=None):
        """sum over ranks (the 1/world factor is applied by the optimizer's grad_scale)."""
        self.check_views()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            host_side = self.grad.is_cuda and dist.get_backend(group) != "nccl"
            if host_side:
                torch.cuda.synchronize()   # gloo stages device tensors through the host: not stream-ordered like RCCL
            dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=group)
            if host_side:
                torch.cuda.synchronize()


class FusedAdamW:
    """torch.optim.AdamW semantics (modules/detection.py:409-441: AdamW, weight_decay 0) on FlatParams,
    one kernel for all parameters; optional clip-by-value (train.py:156-157).  On CPU tensors (gloo
    tests of the data-parallel plumbing) the same update is evaluated with torch ops."""

    def __init__(self, fp: FlatParams, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, clip_value=0.0):
        self.fp, self.betas, self.eps, self.wd, self.clip = fp, betas, eps, weight_decay, clip_value
        self.m = torch.zeros_like(fp.flat)
        self.v = torch.zeros_like(fp.flat)
        self.lr_step = torch.tensor([lr, 0.0], device=fp.flat.device, dtype=torch.float32)
        self._one = torch.tensor([0.0, 1.0], device=fp.flat.device, dtype=torch.float32)

    def set_lr(self, lr: float):
        self.lr_step[0] = lr

    @torch.no_grad()
    def step(self, grad_scale: float = 1.0):
        self.lr_step += self._one
        fp = self.fp
        fp.check_views()
        if fp.flat.is_cuda:
            from . import functional as SF
            SF.adamw_step(fp.flat, fp.grad, self.m, self.v, self.lr_step, self.betas[0], self.betas[1], self.eps, self.wd,
                          grad_scale, self.clip)
            return
        lr, t = float(self.lr_step[0]), float(self.lr_step[1])
        g = fp.grad * grad_scale
        if self.clip > 0:
            g = g.clamp(-self.clip, self.clip)
        b1, b2 = self.betas
        fp.flat.mul_(1 - lr * self.wd)
        self.m.mul_(b1).add_(g, alpha=1 - b1)
        self.v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = self.v.sqrt() / (1 - b2 ** t) ** 0.5 + self.eps
        fp.flat.addcdiv_(self.m, denom, value=-lr / (1 - b1 ** t))
