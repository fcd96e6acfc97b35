"""Data-parallel plumbing: flat parameter / gradient buffers, one gradient all-reduce per step over
RCCL (torch.distributed backend "nccl" on ROCm; "gloo" in CPU tests), fused AdamW on the flat buffers.

The hot path shards over the batch only: every rank runs the full model on its own
samples; the only exchange is the gradient all-reduce (the reference uses DDP, train.py:96-98).
BatchNorm statistics in the PAFPN stay rank-local (documented deviation from the reference's
SyncBatchNorm under DDP, train.py:167; identical to the reference's single-GPU numerics).
"""
from __future__ import annotations

from typing import Iterable, List, Optional

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

    def __init__(self, modules: Iterable[torch.nn.Module], buckets: Optional[List[List[torch.nn.Module]]] = None):
        """buckets: optional partition of the modules into gradient buckets (lists of modules), in the order in which their
        gradients become final during the backward pass; each bucket is a contiguous slice of the flat buffers
        (`bucket_ranges`), so it can be all-reduced and updated while the backward of the later buckets still runs."""
        groups = buckets if buckets is not None else [list(modules)]
        seen, params, bounds = set(), [], []
        for grp in groups:
            start = sum((p.numel() + 3) // 4 * 4 for p in params)
            for m in grp:
                for p in m.parameters():
                    if id(p) not in seen and p.requires_grad:
                        seen.add(id(p))
                        params.append(p)
            bounds.append((start, sum((p.numel() + 3) // 4 * 4 for p in params)))
        self.bucket_ranges = bounds
        self.params: List[torch.nn.Parameter] = params
        n = sum((p.numel() + 3) // 4 * 4 for p in params)
        dev = params[0].device
        self.flat = torch.zeros(n, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(n, device=dev, dtype=torch.float32)
        off = 0
        self.offsets: List[int] = []
        with torch.no_grad():
            for p in params:
                k = p.numel()
                self.offsets.append(off)
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

    def all_reduce(self, group=None, bucket: Optional[int] = None):
        """sum over ranks (the 1/world factor is applied by the optimizer's grad_scale); bucket: only that slice."""
        if bucket is None:
            self.check_views()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            t = self.grad if bucket is None else self.grad[self.bucket_ranges[bucket][0]:self.bucket_ranges[bucket][1]]
            host_side = self.grad.is_cuda and dist.get_backend(group) != "nccl"
            if host_side:
                torch.cuda.synchronize()   # gloo stages device tensors through the host: not stream-ordered like RCCL
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            if host_side:
                torch.cuda.synchronize()


class OneCycleLR:
    """torch.optim.lr_scheduler.OneCycleLR(anneal_strategy='linear', cycle_momentum=False, three_phase=False) as the reference
    configures it (modules/detection.py:418-431: final lr = max_lr / final_div_factor, i.e. torch's final_div_factor is
    final_div_factor / div_factor).  The same closed form is evaluated on the device inside the fused AdamW kernel from its step
    counter (sast_adamw_onecycle), so a replayed hipGraph advances the schedule without host interaction; `lr_at` is the host
    copy (logging, tests)."""

    def __init__(self, max_lr: float, total_steps: int, pct_start: float = 0.005, div_factor: float = 25.0, final_div_factor: float = 10000.0):
        assert total_steps > 0
        self.max_lr, self.total_steps = float(max_lr), int(total_steps)
        self.initial_lr = self.max_lr / div_factor
        self.min_lr = self.initial_lr / (final_div_factor / div_factor)
        self.end1 = float(pct_start * total_steps) - 1.0
        self.end2 = float(total_steps - 1)

    def lr_at(self, step_num: int) -> float:
        """learning rate the optimizer step number step_num + 1 uses (torch: scheduler.last_epoch == step_num)"""
        sn = float(step_num)
        if sn <= self.end1:
            return self.initial_lr + (self.max_lr - self.initial_lr) * (sn / self.end1 if self.end1 > 0 else 1.0)
        return self.max_lr + (self.min_lr - self.max_lr) * ((sn - self.end1) / (self.end2 - self.end1))


class FusedAdamW:
    """torch.optim.AdamW semantics (modules/detection.py:409-441: AdamW, weight_decay 0) on FlatParams,
    one kernel for all parameters; optional clip-by-value (train.py:156-157); optional OneCycleLR schedule evaluated on the
    device.  Every element of the flat buffer is updated: a parameter that received no gradient counts as gradient 0 (torch
    skips grad=None parameters -- identical here, where every parameter is on the loss path).  On CPU tensors (gloo tests of
    the data-parallel plumbing) the same update is evaluated with torch ops."""

    def __init__(self, fp: FlatParams, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, clip_value=0.0, schedule: Optional[OneCycleLR] = None):
        self.fp, self.betas, self.eps, self.wd, self.clip = fp, betas, eps, weight_decay, clip_value
        self.schedule = schedule
        self.m = torch.zeros_like(fp.flat)
        self.v = torch.zeros_like(fp.flat)
        self.lr_step = torch.tensor([lr, 0.0], device=fp.flat.device, dtype=torch.float32)
        self._one = torch.tensor([0.0, 1.0], device=fp.flat.device, dtype=torch.float32)

    def set_lr(self, lr: float):
        self.lr_step[0] = lr

    @torch.no_grad()
    def begin_step(self):
        """advance the step counter (once per optimizer step, before the bucket updates)"""
        self.lr_step += self._one

    @torch.no_grad()
    def update(self, grad_scale: float = 1.0, bucket: Optional[int] = None):
        """the AdamW update of one bucket (or of everything) at the current step count"""
        fp = self.fp
        a, b = (0, fp.numel) if bucket is None else fp.bucket_ranges[bucket]
        if b <= a:
            return
        p, g, m, v = fp.flat[a:b], fp.grad[a:b], self.m[a:b], self.v[a:b]
        if fp.flat.is_cuda:
            from . import functional as SF
            if self.schedule is not None:
                SF.adamw_onecycle_step(p, g, m, v, self.lr_step, self.schedule, self.betas[0], self.betas[1], self.eps, self.wd, grad_scale, self.clip)
            else:
                SF.adamw_step(p, g, m, v, self.lr_step, self.betas[0], self.betas[1], self.eps, self.wd, grad_scale, self.clip)
            return
        t = float(self.lr_step[1])
        lr = self.schedule.lr_at(int(t) - 1) if self.schedule is not None else float(self.lr_step[0])
        g = g * grad_scale
        if self.clip > 0:
            g = g.clamp(-self.clip, self.clip)
        b1, b2 = self.betas
        p.mul_(1 - lr * self.wd)
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = v.sqrt() / (1 - b2 ** t) ** 0.5 + self.eps
        p.addcdiv_(m, denom, value=-lr / (1 - b1 ** t))

    @torch.no_grad()
    def step(self, grad_scale: float = 1.0):
        self.fp.check_views()
        self.begin_step()
        self.update(grad_scale)
