"""Mirror of models/detection/yolox_extension/models/yolo_pafpn.py (YOLOPAFPN) on NHWC rows."""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch.nn as nn

from .. import functional as SF
from .network_blocks import BaseConv, BnArena, CSPLayer, DWConv, bn_scratch_floats, pass_sync_group


class YOLOPAFPN(nn.Module):
    """yolo_pafpn.py:18-139: top-down + bottom-up FPN over backbone stages (2,3,4)."""

    def __init__(self, depth: float = 1.0, in_stages: Tuple[int, ...] = (2, 3, 4), in_channels: Tuple[int, ...] = (256, 512, 1024),
                 depthwise: bool = False, act: str = "silu", compile_cfg: Optional[Dict] = None):
        super().__init__()
        assert len(in_stages) == len(in_channels) == 3, 'Current implementation only for 3 feature maps'
        if compile_cfg is not None and compile_cfg.get('enable', False):
            # compile.enable asks the reference for torch.compile (sast_rnn.py:86-95 / yolo_pafpn.py:40-45): an optimisation, not a change
            # of results.  There is no tracing compiler here -- the forward is hand-written launches and the step replays as hipGraphs
            import warnings
            warnings.warn("sast_amd: compile.enable is ignored (no torch.compile: capture the step in a hipGraph, sast_amd.training.TrainStep)")
        self.in_features, self.in_channels = in_stages, in_channels
        c0, c1, c2 = in_channels
        n = round(3 * depth)
        Conv = DWConv if depthwise else BaseConv      # yolo_pafpn.py:37: the two bottom-up convs (and every Bottleneck.conv2)
        self.lateral_conv0 = BaseConv(c2, c1, 1, 1, act=act)
        self.C3_p4 = CSPLayer(2 * c1, c1, n, False, depthwise=depthwise, act=act)
        self.reduce_conv1 = BaseConv(c1, c0, 1, 1, act=act)
        self.C3_p3 = CSPLayer(2 * c0, c0, n, False, depthwise=depthwise, act=act)
        self.bu_conv2 = Conv(c0, c0, 3, 2, act=act)
        self.C3_n3 = CSPLayer(2 * c0, c1, n, False, depthwise=depthwise, act=act)
        self.bu_conv1 = Conv(c1, c1, 3, 2, act=act)
        self.C3_n4 = CSPLayer(2 * c1, c2, n, False, depthwise=depthwise, act=act)
        self._sync_group = None      # set by network_blocks.convert_sync_batchnorm

    def forward_nhwc(self, feats: Dict[int, object]):
        x2, x1, x0 = (feats[f] for f in self.in_features)
        if not hasattr(self, "_bn_floats"):
            self._bn_floats = bn_scratch_floats(self)
        ar = BnArena(self._bn_floats, x0.device)                            # one memset for all 32 BatchNorm reductions
        grp = pass_sync_group(self) if self.training else None
        token = None
        if grp is not None and grp.active():
            token = grp.exchange_batch(x0.shape[0], x0.device)                # SyncBatchNorm: the sample counts of all ranks, once per pass
        # outputs with two consumers come as (y, alias) pairs: autograd then delivers the two gradients separately and the producing
        # conv's BatchNorm-backward kernels add them while reading (no accumulation launch)
        fpn_out0, fpn_out0b = self.lateral_conv0.forward_nhwc(x0, ar, two_outputs=True)
        f_out0 = self.C3_p4.forward_nhwc(SF.upsample_cat(fpn_out0, x1), ar, sole_input=True)  # nearest-exact x2 + cat (yolo_pafpn.py:118-121)
        fpn_out1, fpn_out1b = self.reduce_conv1.forward_nhwc(f_out0, ar, sole=True, two_outputs=True)      # f_out0 has no other consumer
        pan_out2, pan_out2b = self.C3_p3.forward_nhwc(SF.upsample_cat(fpn_out1, x2), ar, sole_input=True, two_outputs=True)
        p_out1 = (self.bu_conv2.forward_nhwc(pan_out2b, ar), fpn_out1b)   # th.cat (yolo_pafpn.py:129) read in place by C3_n3's 1x1 convs
        pan_out1, pan_out1b = self.C3_n3.forward_nhwc(p_out1, ar, sole_input=(True, False), two_outputs=True)   # the bottom-up conv feeds only this layer
        p_out0 = (self.bu_conv1.forward_nhwc(pan_out1b, ar), fpn_out0b)   # th.cat (yolo_pafpn.py:134)
        pan_out0 = self.C3_n4.forward_nhwc(p_out0, ar, sole_input=(True, False))
        ar.finish()
        if token is not None:          # a head handed these very tensors takes the exchange over (SyncBatchNormGroup.same_pass)
            pan_out2._sast_sync_pass = pan_out1._sast_sync_pass = pan_out0._sast_sync_pass = token
        return pan_out2, pan_out1, pan_out0

    def forward(self, input):
        """input: {stage: (B,C,H,W)} -> 3-tuple of (B,C,H,W) maps (channels-last memory)."""
        feats = {f: SF.as_nhwc(input[f]) for f in self.in_features}
        return tuple(SF.as_nchw_view(o) for o in self.forward_nhwc(feats))
