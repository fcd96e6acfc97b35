"""YOLOX head, INFERENCE path (reference models/detection/yolox/models/yolo_head.py:21-289).

Same constructor, sub-module names and state_dict keys as the reference `YOLOXHead`, so a reference checkpoint loads with
`strict=True`.  `forward(xin)` in eval mode returns `(outputs, None)` with outputs (B, n_anchors_all, 5 + num_classes), decoded
like `decode_outputs` when `decode_in_inference` is set.  The 15 Conv+BN+SiLU units run through the same HIP op as the PAFPN
(`sast_conv_bn_silu_fwd`, running statistics), the three 1x1 prediction convs + sigmoid + box decode of a level are one kernel
(`sast_head_pred_fwd`).  The TRAINING branch (yolo_head.py:291-606) runs the SimOTA assignment and the IoU / objectness /
class losses on the device for the whole batch without a host sync (`sast_yolox_loss`); `use_l1` (off by default in the
reference) is supported."""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn as nn

from .. import _lib as L
from .. import functional as SF
from .network_blocks import BaseConv, BnArena, DWConv, bn_scratch_floats, forward_sync_group, pass_sync_group


class _PredConv(nn.Module):
    """parameter holder with the names / shapes of the reference's nn.Conv2d(hidden, out, 1) (yolo_head.py:104-133)."""

    def __init__(self, cin: int, cout: int):
        super().__init__()
        k = 1.0 / math.sqrt(cin)
        self.weight = nn.Parameter((torch.rand(cout, cin, 1, 1) * 2 - 1) * k)
        self.bias = nn.Parameter((torch.rand(cout) * 2 - 1) * k)


class YOLOXHead(nn.Module):
    def __init__(self, num_classes=80, strides=(8, 16, 32), in_channels=(256, 512, 1024), act="silu", depthwise=False,
                 compile_cfg: Optional[Dict] = None):
        super().__init__()
        Conv = DWConv if depthwise else BaseConv      # yolo_head.py:42: the 3x3 tower convs
        self.depthwise = bool(depthwise)
        self.num_classes = num_classes
        self.decode_in_inference = True
        self.strides = tuple(strides)
        hidden = int(256 * (in_channels[-1] / 1024))          # yolo_head.py:48-56
        self.hidden_dim = hidden
        self.cls_convs, self.reg_convs = nn.ModuleList(), nn.ModuleList()
        self.cls_preds, self.reg_preds, self.obj_preds, self.stems = nn.ModuleList(), nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        for c in in_channels:
            self.stems.append(BaseConv(int(c), hidden, 1, stride=1, act=act))
            self.cls_convs.append(nn.Sequential(Conv(hidden, hidden, 3, stride=1, act=act), Conv(hidden, hidden, 3, stride=1, act=act)))
            self.reg_convs.append(nn.Sequential(Conv(hidden, hidden, 3, stride=1, act=act), Conv(hidden, hidden, 3, stride=1, act=act)))
            self.cls_preds.append(_PredConv(hidden, num_classes))
            self.reg_preds.append(_PredConv(hidden, 4))
            self.obj_preds.append(_PredConv(hidden, 1))
        self.use_l1 = False
        self._sync_group = None      # set by network_blocks.convert_sync_batchnorm
        self.hw = None
        self.initialize_biases(prior_prob=0.01)

    def initialize_biases(self, prior_prob):          # yolo_head.py:154-163
        with torch.no_grad():
            for conv in list(self.cls_preds) + list(self.obj_preds):
                conv.bias.fill_(-math.log((1 - prior_prob) / prior_prob))

    def forward_train_nhwc(self, feats, labels):
        """training branch: feats three (B,H,W,C) NHWC maps (autograd tensors), labels (B, max_labels, 5) = (cls, cx, cy, w, h)
        -> (predictions (B, A, 5+nc), losses dict like yolo_head.py:224-231)"""
        per_level, levels = [], []
        if not hasattr(self, "_bn_floats"):
            self._bn_floats = bn_scratch_floats(self)
        ar = BnArena(self._bn_floats, feats[0].device)     # one memset / one counter update for the 15 BatchNorms
        grp = pass_sync_group(self) if self.training else None
        sync = grp is not None and grp.active()
        if sync and not grp.same_pass(feats[0]):
            grp.exchange_batch(feats[0].shape[0], feats[0].device)
        if sync and SF.SYNC_BN_GROUPS and not self.depthwise:
            # SyncBatchNorm: the three levels are independent, and so are the two towers of a level -- the 15 units run as three sets
            # (stems; first 3x3 of both towers; second 3x3) with ONE statistics all-reduce per set and direction instead of 15
            n = len(feats)
            xs = forward_sync_group(grp, list(self.stems), list(feats), ar)
            t = forward_sync_group(grp, [self.cls_convs[k][0] for k in range(n)] + [self.reg_convs[k][0] for k in range(n)], xs + xs, ar)
            t = forward_sync_group(grp, [self.cls_convs[k][1] for k in range(n)] + [self.reg_convs[k][1] for k in range(n)], t, ar, sole=True)
            for k, (x, stride) in enumerate(zip(xs, self.strides)):
                cp, rp, op = self.cls_preds[k], self.reg_preds[k], self.obj_preds[k]
                per_level.append((t[n + k], t[k], rp.weight, rp.bias, op.weight, op.bias, cp.weight, cp.bias))
                levels.append((x.shape[1], x.shape[2], stride))
        else:
            for k, (x, stride) in enumerate(zip(feats, self.strides)):
                x = self.stems[k].forward_nhwc(x, ar)
                if SF.CONV_PAIR and not sync and not self.depthwise:
                    # the first conv of both towers reads the stem output: one stacked 3x3 GEMM + shared BatchNorm launches, and (sole
                    # consumer of the stem output) the stem's BatchNorm-backward reduction in the pair's dX epilogue
                    c0, r0 = self.cls_convs[k][0], self.reg_convs[k][0]
                    args = [(c.conv.weight, c.bn.weight, c.bn.bias, c.bn.running_mean, c.bn.running_var, c.bn.momentum, c.bn.eps) for c in (c0, r0)]
                    ws = tuple(ar.take(SF.bn_ws_floats(c.bn.num_features)) for c in (c0, r0))
                    cf, rf = SF.conv_bn_silu2(x, args[0], args[1], ws, sole_consumer=True, ksize=3)
                    ar.counters.extend(c.bn.num_batches_tracked for c in (c0, r0) if c.bn.num_batches_tracked is not None)
                    first = 1
                else:
                    cf, rf = x, x          # the stem output feeds both towers; inside a tower every conv has one consumer
                    first = 0
                for i, conv in enumerate(self.cls_convs[k]):
                    if i >= first:
                        cf = conv.forward_nhwc(cf, ar, sole=i > 0)
                for i, conv in enumerate(self.reg_convs[k]):
                    if i >= first:
                        rf = conv.forward_nhwc(rf, ar, sole=i > 0)
                cp, rp, op = self.cls_preds[k], self.reg_preds[k], self.obj_preds[k]
                per_level.append((rf, cf, rp.weight, rp.bias, op.weight, op.bias, cp.weight, cp.bias))
                levels.append((x.shape[1], x.shape[2], stride))
        ar.finish()
        losses, pred, fg, mg, piou = SF.head_pred_loss(labels, levels, self.num_classes, self.decode_in_inference, per_level, self.use_l1)
        self.last_assignment = (fg, mg, piou)      # SimOTA result of this step (device tensors), for inspection / tests
        self.hw = [(h, w) for h, w, _ in levels]
        det = losses.detach()
        return pred, {"loss": losses[0], "iou_loss": det[1], "conf_loss": det[2], "cls_loss": det[3], "l1_loss": det[4] if self.use_l1 else 0.0,
                      "num_fg": det[5]}

    @torch.no_grad()
    def forward_nhwc(self, feats):
        """inference: feats three (B,H,W,C) NHWC maps -> (B, n_anchors_all, 5 + num_classes)"""
        if self.training:
            raise RuntimeError("sast_amd: YOLOXHead.forward_nhwc is the inference path; in training mode call forward(xin, labels)")
        B = feats[0].shape[0]
        dev = feats[0].device
        hw = [tuple(f.shape[1:3]) for f in feats]
        A = sum(h * w for h, w in hw)
        no = 5 + self.num_classes
        out = torch.empty(B, A, no, device=dev)
        off = 0
        for k, (x, stride) in enumerate(zip(feats, self.strides)):
            x = self.stems[k].forward_nhwc(x)
            if SF.CONV_PAIR and not self.depthwise:     # the first conv of both towers reads the stem output: one launch over the stacked 3x3 weights
                c0, r0 = self.cls_convs[k][0], self.reg_convs[k][0]
                cf, rf = SF.conv_bn_silu2_infer(x, *[(c.conv.weight, c.bn.weight, c.bn.bias, c.bn.running_mean, c.bn.running_var,
                                                     c.bn.momentum, c.bn.eps) for c in (c0, r0)], ksize=3)
                for conv in list(self.cls_convs[k])[1:]:
                    cf = conv.forward_nhwc(cf)
                for conv in list(self.reg_convs[k])[1:]:
                    rf = conv.forward_nhwc(rf)
            else:
                cf, rf = x, x
                for conv in self.cls_convs[k]:
                    cf = conv.forward_nhwc(cf)
                for conv in self.reg_convs[k]:
                    rf = conv.forward_nhwc(rf)
            h, w = hw[k]
            cp, rp, op = self.cls_preds[k], self.reg_preds[k], self.obj_preds[k]
            L.check(L.lib().sast_head_pred_decode(rf.data_ptr(), cf.data_ptr(), rp.weight.data_ptr(), rp.bias.data_ptr(), op.weight.data_ptr(),
                                                  op.bias.data_ptr(), cp.weight.data_ptr(), cp.bias.data_ptr(), out.data_ptr(), B, h, w,
                                                  self.hidden_dim, self.num_classes, float(stride), off, A, int(self.decode_in_inference),
                                                  SF._stream()), "head_pred_decode")
            off += h * w
        self.hw = hw
        return out

    def forward(self, xin, labels=None):
        """xin: the three PAFPN maps (B,C,H,W) -> (outputs, losses)   (yolo_head.py:165-246; losses is None in eval mode)"""
        SF._need_gpu(*xin)
        feats = [SF.as_nhwc(x.float()) for x in xin]
        if self.training:
            if labels is None:
                raise ValueError("sast_amd: YOLOXHead in training mode needs labels (yolo_head.py:215-231)")
            return self.forward_train_nhwc(feats, labels)
        return self.forward_nhwc(feats), None
