"""Mirror of models/detection/yolox/models/network_blocks.py (BaseConv / Bottleneck / CSPLayer) on NHWC rows."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import functional as SF
from ..layers.ops import channels_last_conv_weight


class BaseConv(nn.Module):
    """Conv2d(no bias, same pad) -> BatchNorm2d -> SiLU (network_blocks.py:29-54) as one fused op."""

    def __init__(self, in_channels, out_channels, ksize, stride, groups=1, bias=False, act="silu"):
        super().__init__()
        depthwise = groups > 1 and groups == in_channels == out_channels      # DWConv.dconv (network_blocks.py:63-70)
        if (groups != 1 and not depthwise) or bias or act != "silu" or ksize not in (1, 3):
            raise NotImplementedError("sast_amd: BaseConv implements groups=1 or groups=in=out (depth-wise), bias=False, act='silu', "
                                      "ksize in {1,3}")
        self.ksize, self.stride, self.groups = ksize, stride, groups
        self.conv = nn.Module()
        self.conv.weight = channels_last_conv_weight(out_channels, in_channels // groups, ksize)
        self.bn = nn.BatchNorm2d(out_channels)
        self.sync_bn = None       # a functional.SyncBatchNormGroup after convert_sync_batchnorm: batch statistics over all ranks

    def sync_group(self):
        """the SyncBatchNormGroup of this unit, or None.  `torch.nn.SyncBatchNorm.convert_sync_batchnorm` (what the reference's
        `Trainer(sync_batchnorm=True)` runs under DDP, train.py:167) replaces `self.bn` by a torch.nn.SyncBatchNorm holding the same
        parameters and buffers; the fused op never calls that module, so its class is the request: statistics over the ranks of its
        `process_group` -- never silently rank-local."""
        if self.sync_bn is None and isinstance(self.bn, nn.SyncBatchNorm):
            self.sync_bn = SF.sync_group_for(self.bn.process_group)
        return self.sync_bn

    def forward_nhwc(self, x, arena=None, sole=False, two_outputs=False):
        """arena: optional `BnArena` handing out zero-filled reduction scratch (one memset per FPN forward) and
        batching the num_batches_tracked increments.  sole: the caller guarantees this conv is the only consumer of x
        (functional.conv_bn_silu: the producing conv's BatchNorm-backward reduction then rides on this conv's dX epilogue).
        two_outputs: return (y, y_alias) for an output that has two consumers (their gradients are then added inside this
        conv's BatchNorm-backward kernels, not by an autograd launch)."""
        bn = self.bn
        ws = arena.take(SF.bn_ws_floats(bn.num_features)) if arena is not None else None
        sync = self.sync_group()
        if sync is not None and arena is None and self.training and sync.active():
            # a unit called on its own (no PAFPN / head pass around it that exchanged the sample counts): its own exchange
            x0 = x[0] if isinstance(x, (tuple, list)) else x
            sync.exchange_batch(x0.shape[0], x0.device)
        y = SF.conv_bn_silu(x, self.conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, self.ksize, self.stride,
                            self.training, bn.momentum, bn.eps, ws, sole_consumer=sole, two_outputs=two_outputs, sync=sync)
        if self.training and bn.num_batches_tracked is not None:
            if arena is not None:
                arena.counters.append(bn.num_batches_tracked)
            else:
                bn.num_batches_tracked.add_(1)
        return y

    def sync_item(self, x, arena=None, sole=False, two_outputs=False):
        """this unit's entry of a `forward_sync_group` call (same arguments as forward_nhwc)"""
        bn = self.bn
        return dict(x_nhwc=x, w=self.conv.weight, bn_w=bn.weight, bn_b=bn.bias, run_mean=bn.running_mean, run_var=bn.running_var,
                    ksize=self.ksize, stride=self.stride, momentum=bn.momentum, eps=bn.eps, sole_consumer=sole, two_outputs=two_outputs,
                    bn_ws=arena.take(SF.bn_ws_floats(bn.num_features)) if arena is not None else None)

    def forward(self, x):
        if isinstance(x, (tuple, list)):
            raise TypeError("sast_amd: the two-source (virtual concat) input is an internal NHWC feature; use forward_nhwc")
        return SF.as_nchw_view(self.forward_nhwc(SF.as_nhwc(x)))


def forward_sync_group(grp, convs, xs, arena, sole=False):
    """several INDEPENDENT BaseConv units under SyncBatchNorm (training mode, `grp.active()`, the pass exchanged its sample counts) with
    ONE statistics all-reduce per direction for all of them (functional.conv_bn_silu_sync_group): the dependent chain of a multi-rank
    step counts collectives, not units.  -> one output per unit"""
    soles = sole if isinstance(sole, (tuple, list)) else (sole,) * len(convs)
    if arena is None:
        # units called on their own (a stand-alone CSPLayer.forward on a converted model: no PAFPN / head pass around them exchanged the
        # sample counts of this batch): their own exchange, as BaseConv.forward_nhwc does -- never a stale ratio of an earlier pass
        x0 = xs[0][0] if isinstance(xs[0], (tuple, list)) else xs[0]
        grp.exchange_batch(x0.shape[0], x0.device)
    ys = SF.conv_bn_silu_sync_group(grp, [c.sync_item(x, arena, sole=s) for c, x, s in zip(convs, xs, soles)])
    for c in convs:
        if c.bn.num_batches_tracked is not None:
            if arena is not None:
                arena.counters.append(c.bn.num_batches_tracked)
            else:
                c.bn.num_batches_tracked.add_(1)
    return ys


class DWConv(nn.Module):
    """network_blocks.py:57-76: a depth-wise k x k unit (groups = channels; carries the stride) followed by a point-wise 1x1 unit, each
    with its own BatchNorm + SiLU.  Parameter names (`dconv.*`, `pconv.*`) are the reference's, so its checkpoints load."""

    def __init__(self, in_channels, out_channels, ksize, stride=1, act="silu"):
        super().__init__()
        self.dconv = BaseConv(in_channels, in_channels, ksize=ksize, stride=stride, groups=in_channels, act=act)
        self.pconv = BaseConv(in_channels, out_channels, ksize=1, stride=1, groups=1, act=act)

    def forward_nhwc(self, x, arena=None, sole=False, two_outputs=False):
        """same contract as BaseConv.forward_nhwc; the depth-wise stencil has no dX epilogue, so a producer of x keeps its own
        BatchNorm-backward reduction (`sole` is not passed on), while the point-wise conv is the only consumer of the stencil's output"""
        return self.pconv.forward_nhwc(self.dconv.forward_nhwc(x, arena), arena, sole=True, two_outputs=two_outputs)

    def forward(self, x):
        return SF.as_nchw_view(self.forward_nhwc(SF.as_nhwc(x)))


def sync_active(conv: "BaseConv") -> bool:
    """training-mode statistics of this unit span several ranks: the stacked two-conv launches (one process's rows) are not used"""
    grp = conv.sync_group()
    return grp is not None and grp.active()


def bn_scratch_floats(module: nn.Module) -> int:
    """BatchNorm reduction scratch of every conv + BatchNorm unit below `module` (BatchNorm2d, or the torch.nn.SyncBatchNorm that
    torch's conversion put in its place)"""
    return sum(SF.bn_ws_floats(m.num_features) for m in module.modules() if isinstance(m, nn.modules.batchnorm._BatchNorm))


def pass_sync_group(module: nn.Module):
    """the SyncBatchNormGroup a PAFPN / head pass exchanges its sample counts on, or None: the one `convert_sync_batchnorm` installed
    (`module._sync_group`), else the one behind torch.nn.SyncBatchNorm modules left by torch's own conversion -- looked up when the
    module's first unit turns out to be converted (the conversion replaces every BatchNorm below a module, so one unit tells) and
    then installed on every unit below `module`."""
    grp = getattr(module, "_sync_group", None)       # (a custom fpn / head without the attribute: nothing installed, torch's conversion below)
    if grp is None:
        units = [m for m in module.modules() if isinstance(m, BaseConv)]
        if any(isinstance(m.bn, nn.SyncBatchNorm) for m in units):       # ANY converted unit: a partial conversion raises below, whichever unit comes first
            groups = {id(m.bn.process_group): m.bn.process_group for m in module.modules()
                      if isinstance(m, BaseConv) and isinstance(m.bn, nn.SyncBatchNorm)}
            if len(groups) != 1:
                raise RuntimeError("sast_amd: the SyncBatchNorm modules of one PAFPN / head span several process groups")
            grp = SF.sync_group_for(next(iter(groups.values())))
            for m in module.modules():
                if isinstance(m, BaseConv):
                    if not isinstance(m.bn, nn.SyncBatchNorm):
                        raise RuntimeError("sast_amd: only some BatchNorm modules of this PAFPN / head were converted to SyncBatchNorm")
                    m.sync_bn = grp
            if hasattr(module, "_sync_group"):
                module._sync_group = grp
    return grp


def convert_sync_batchnorm(module: nn.Module, process_group=None, force: bool = False):
    """the reference's Trainer(sync_batchnorm=True) under DDP (train.py:167; torch.nn.SyncBatchNorm.convert_sync_batchnorm): every
    conv + BatchNorm + SiLU unit below `module` takes its training-mode batch statistics over the rows of ALL ranks of `process_group`.
    The BatchNorm2d modules stay what they are (same parameters, buffers and state_dict keys); one functional.SyncBatchNormGroup is shared
    by the units and issues the all-reduces between the two phases of the fused op.  `force`: keep the two-phase path on with one rank
    (functional.SyncBatchNormGroup).  Returns `module`."""
    grp = SF.SyncBatchNormGroup(process_group, force=force)
    for m in module.modules():
        if isinstance(m, BaseConv):
            m.sync_bn = grp
        if hasattr(m, "_sync_group"):
            m._sync_group = grp
    return module


class BnArena:
    """zero-filled scratch for the BatchNorm reductions of a whole FPN pass: ONE memset instead of one per conv."""

    def __init__(self, n_floats, device):
        self.buf = torch.zeros(n_floats, device=device)
        self.off = 0
        self.counters = []

    def take(self, n):
        if self.off + n > self.buf.numel():
            return None
        t = self.buf[self.off:self.off + n]
        self.off += n
        return t

    def finish(self):
        if self.counters:
            torch._foreach_add_(self.counters, 1)
            self.counters = []


class Bottleneck(nn.Module):
    """network_blocks.py:79-101"""

    def __init__(self, in_channels, out_channels, shortcut=True, expansion=0.5, depthwise=False, act="silu"):
        super().__init__()
        hidden = int(out_channels * expansion)
        Conv = DWConv if depthwise else BaseConv
        self.conv1 = BaseConv(in_channels, hidden, 1, stride=1, act=act)
        self.conv2 = Conv(hidden, out_channels, 3, stride=1, act=act)
        self.use_add = shortcut and in_channels == out_channels

    def forward_nhwc(self, x, arena=None, sole_input=False):
        # sole_input: x has no consumer besides this block (with the shortcut the add is a second one)
        y = self.conv2.forward_nhwc(self.conv1.forward_nhwc(x, arena, sole=sole_input and not self.use_add), arena, sole=True)
        return y + x if self.use_add else y

    def forward(self, x):
        return SF.as_nchw_view(self.forward_nhwc(SF.as_nhwc(x)))


class CSPLayer(nn.Module):
    """network_blocks.py:104-141"""

    def __init__(self, in_channels, out_channels, n=1, shortcut=True, expansion=0.5, depthwise=False, act="silu"):
        super().__init__()
        hidden = int(out_channels * expansion)
        self.conv1 = BaseConv(in_channels, hidden, 1, stride=1, act=act)
        self.conv2 = BaseConv(in_channels, hidden, 1, stride=1, act=act)
        self.conv3 = BaseConv(2 * hidden, out_channels, 1, stride=1, act=act)
        self.m = nn.Sequential(*[Bottleneck(hidden, hidden, shortcut, 1.0, depthwise, act=act) for _ in range(n)])

    def _pair_args(self):
        return tuple((c.conv.weight, c.bn.weight, c.bn.bias, c.bn.running_mean, c.bn.running_var, c.bn.momentum, c.bn.eps)
                     for c in (self.conv1, self.conv2))

    def _conv12(self, x, arena, sole_input):
        cs = (self.conv1, self.conv2)
        ws = tuple(arena.take(SF.bn_ws_floats(c.bn.num_features)) if arena is not None else None for c in cs)
        args = self._pair_args()
        ys = SF.conv_bn_silu2(x, args[0], args[1], ws, sole_consumer=sole_input)
        for c in cs:
            if c.bn.num_batches_tracked is not None:
                if arena is not None:
                    arena.counters.append(c.bn.num_batches_tracked)
                else:
                    c.bn.num_batches_tracked.add_(1)
        return ys

    def forward_nhwc(self, x, arena=None, sole_input=False, two_outputs=False):
        """sole_input: nothing but this layer consumes x (then the BatchNorm-backward reductions of the convs that produced x
        ride on this layer's input-gradient launch); two_outputs: see BaseConv.forward_nhwc"""
        if self.training and SF.CONV_PAIR and not sync_active(self.conv1):
            # conv1 and conv2 read the same input: one GEMM over the stacked weights, one BatchNorm pass for both, and a
            # backward whose dX is already the sum of the two input gradients
            x1, x2 = self._conv12(x, arena, sole_input)
        elif not self.training and SF.CONV_PAIR and not torch.is_grad_enabled():
            x1, x2 = SF.conv_bn_silu2_infer(x, self._pair_args()[0], self._pair_args()[1])     # inference: one launch for both
        elif self.training and sync_active(self.conv1) and SF.SYNC_BN_GROUPS:
            # SyncBatchNorm: the stacked launch sees one process's rows only; the two units still share ONE statistics all-reduce
            x1, x2 = forward_sync_group(self.conv1.sync_group(), (self.conv1, self.conv2), (x, x), arena)
        else:
            x1 = self.conv1.forward_nhwc(x, arena)
            x2 = self.conv2.forward_nhwc(x, arena)
        for b in self.m:
            x1 = b.forward_nhwc(x1, arena, sole_input=True)
        return self.conv3.forward_nhwc((x1, x2), arena, sole=True, two_outputs=two_outputs)   # th.cat((x_1, x_2)) read in place by the 1x1 conv

    def forward(self, x):
        return SF.as_nchw_view(self.forward_nhwc(SF.as_nhwc(x)))
