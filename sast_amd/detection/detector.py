"""models/detection/yolox_extension/models/detector.py:18-80 -- backbone + PAFPN + YOLOX head behind the reference's
`YoloXDetector` surface (inference and training branch)."""
from __future__ import annotations

from typing import Optional

import torch.nn as nn

from ..layers.ops import cfg_get
from .sast_rnn import RNNDetector
from .yolo_head import YOLOXHead
from .yolo_pafpn import YOLOPAFPN


class YoloXDetector(nn.Module):
    def __init__(self, model_cfg):
        super().__init__()
        backbone_cfg, fpn_cfg, head_cfg = model_cfg.backbone, model_cfg.fpn, model_cfg.head
        if backbone_cfg.name != 'SASTRNN':
            raise NotImplementedError
        self.backbone = RNNDetector(backbone_cfg)
        in_stages = tuple(fpn_cfg.in_stages)
        in_channels = self.backbone.get_stage_dims(in_stages)
        self.fpn = YOLOPAFPN(depth=fpn_cfg.depth, in_stages=in_stages, in_channels=in_channels,
                             depthwise=cfg_get(fpn_cfg, 'depthwise', False), act=cfg_get(fpn_cfg, 'act', 'silu'))
        strides = self.backbone.get_strides(in_stages)
        self.yolox_head = YOLOXHead(num_classes=head_cfg.num_classes, strides=strides, in_channels=in_channels,
                                    act=cfg_get(head_cfg, 'act', 'silu'), depthwise=cfg_get(head_cfg, 'depthwise', False))

    def forward_backbone(self, x, previous_states=None, token_mask=None):
        return self.backbone(x, previous_states, token_mask)

    def forward_detect(self, backbone_features, targets: Optional[object] = None):
        fpn_features = self.fpn(backbone_features)
        if self.training:
            assert targets is not None
            return self.yolox_head(fpn_features, targets)
        outputs, losses = self.yolox_head(fpn_features)
        assert losses is None
        return outputs, losses

    def forward(self, x, previous_states=None, retrieve_detections: bool = True, targets=None):
        backbone_features, states, p = self.forward_backbone(x, previous_states)
        if not retrieve_detections:
            assert targets is None
            return None, None, states
        outputs, losses = self.forward_detect(backbone_features=backbone_features, targets=targets)
        return outputs, losses, states, p
