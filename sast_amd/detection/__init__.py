from .sast_rnn import RNNDetector, RNNDetectorStage, SASTAttentionPairCl, PositionEmbeddingSine, non_zero_ratio  # noqa: F401
from .yolo_pafpn import YOLOPAFPN  # noqa: F401
from .network_blocks import BaseConv, Bottleneck, CSPLayer, DWConv, convert_sync_batchnorm  # noqa: F401
from .yolo_head import YOLOXHead  # noqa: F401
from .detector import YoloXDetector  # noqa: F401
from ..functional import postprocess  # noqa: F401  (models/detection/yolox/utils/boxes.py:32-76)
from .sequence import BackboneFeatureSelector, RNNStates  # noqa: F401


def build_recurrent_backbone(backbone_cfg):
    """models/detection/recurrent_backbone/__init__.py:6"""
    if backbone_cfg.name == 'SASTRNN':
        return RNNDetector(backbone_cfg)
    raise NotImplementedError
