"""Import the upstream reference (Peterande/SAST) in THIS container only.

Used by `make_golden.py` (fixture generation) and by the container-only
cross-checks in tests/ that are skipped when /root/reference is absent
(it never exists on the GPU box).  Nothing from the reference is copied:
we put three stub modules in sys.modules (omegaconf, strenum, torchvision)
because the reference imports them at module scope (sast_rnn.py:13,21 ->
data/utils/types.py:3-6; yolox/utils/boxes.py:8) and they are not installed
here.
"""
import enum
import os
import sys
import types

REF_ROOT = os.environ.get("SAST_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "models", "layers", "SAST"))


class DictConfig(dict):
    """attribute-access dict standing in for omegaconf.DictConfig."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return v

    def __setattr__(self, k, v):
        self[k] = v


def to_cfg(d):
    if isinstance(d, dict):
        return DictConfig({k: to_cfg(v) for k, v in d.items()})
    if isinstance(d, (list, tuple)):
        return type(d)(to_cfg(v) for v in d)
    return d


def _install_stubs():
    if "omegaconf" not in sys.modules:
        m = types.ModuleType("omegaconf")
        m.DictConfig = DictConfig

        class OmegaConf:
            @staticmethod
            def to_container(cfg, resolve=True, throw_on_missing=True):
                def conv(x):
                    if isinstance(x, dict):
                        return {k: conv(v) for k, v in x.items()}
                    if isinstance(x, (list, tuple)):
                        return [conv(v) for v in x]
                    return x
                return conv(cfg)

        m.OmegaConf = OmegaConf
        m.open_dict = lambda cfg: cfg
        sys.modules["omegaconf"] = m
    if "torchvision" not in sys.modules:   # models/detection/yolox/utils/boxes.py:8 imports it at module scope (NMS only)
        m = types.ModuleType("torchvision")
        m.ops = types.ModuleType("torchvision.ops")
        sys.modules["torchvision"] = m
        sys.modules["torchvision.ops"] = m.ops
    if "strenum" not in sys.modules:
        m = types.ModuleType("strenum")

        class StrEnum(str, enum.Enum):
            pass

        m.StrEnum = StrEnum
        sys.modules["strenum"] = m


def import_reference():
    """Returns a namespace with the reference classes on the hot path."""
    if not reference_available():
        raise RuntimeError(f"reference not found under {REF_ROOT}")
    sys.dont_write_bytecode = True  # never drop __pycache__ into /root/reference
    _install_stubs()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    ns = types.SimpleNamespace()
    from models.detection.recurrent_backbone import sast_rnn as _rnn
    from models.layers.SAST import SAST as _sast
    from models.layers.SAST import ops as _ops
    from models.layers import rnn as _lstm
    from models.detection.yolox_extension.models import yolo_pafpn as _fpn
    ns.sast_rnn, ns.SAST, ns.ops, ns.rnn, ns.yolo_pafpn = _rnn, _sast, _ops, _lstm, _fpn
    try:
        from models.detection.yolox.models import yolo_head as _head
        ns.yolo_head = _head
    except Exception as e:   # noqa: BLE001  (the head is a "next" row: fixtures for it are optional)
        ns.yolo_head = None
        ns.yolo_head_error = repr(e)
    return ns


def backbone_cfg(in_res_hw, partition_size, embed_dim=64, num_blocks=(1, 1, 1, 1),
                 amp=2e-4, bounce=1e-3, ls_init=1e-5, enable_cb=False, input_channels=20):
    """mirror of config/model/sast_yolox/default.yaml after config/modifier.py."""
    return to_cfg({
        "name": "SASTRNN", "compile": None, "input_channels": input_channels,
        "enable_masking": False, "partition_split_32": 2, "embed_dim": embed_dim,
        "dim_multiplier": [1, 2, 4, 8], "num_blocks": list(num_blocks),
        "T_max_chrono_init": [4, 8, 16, 32], "stem": {"patch_size": 4},
        "in_res_hw": tuple(in_res_hw),
        "stage": {
            "downsample": {"type": "patch", "overlap": True, "norm_affine": True},
            "attention": {
                "use_torch_mha": False, "partition_size": tuple(partition_size), "dim_head": 32,
                "attention_bias": True, "mlp_activation": "gelu", "mlp_gated": False, "mlp_bias": True,
                "mlp_ratio": 4, "drop_mlp": 0, "drop_path": 0, "ls_init_value": ls_init,
                "enable_CB": enable_cb, "AMP": amp, "BOUNCE": bounce},
            "lstm": {"dws_conv": False, "dws_conv_only_hidden": True, "dws_conv_kernel_size": 3,
                     "drop_cell_update": 0},
        },
    })
