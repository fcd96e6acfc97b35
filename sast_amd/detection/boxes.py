"""models/detection/yolox/utils/boxes.py:32-76 -- `postprocess` (confidence filter + class-aware NMS) on the device.
The reference delegates the NMS to torchvision.ops.batched_nms; here it is `sast_postprocess` (csrc/k_head.hip)."""
from ..functional import postprocess  # noqa: F401
