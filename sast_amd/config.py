"""Model configuration mirrors: config/model/sast_yolox/default.yaml after config/modifier.py:27-47 has injected
`in_res_hw` and `partition_size`.  The reference uses hydra/omegaconf; any attribute-access mapping works here."""
from __future__ import annotations

import math


class AttrDict(dict):
    """dict with attribute access (stands in for omegaconf.DictConfig)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def to_attr(d):
    if isinstance(d, dict):
        return AttrDict({k: to_attr(v) for k, v in d.items()})
    if isinstance(d, (list, tuple)):
        return type(d)(to_attr(v) for v in d)
    return d


def modified_hw(dataset_hw, partition_split_32=2):
    """config/modifier.py:27-41: pad (H,W) up to a multiple of 32*split and derive the partition size."""
    mult = 32 * partition_split_32
    hw = tuple(math.ceil(x / mult) * mult for x in dataset_hw)
    return hw, tuple(x // mult for x in hw)


def backbone_config(in_res_hw, partition_size, embed_dim=64, num_blocks=(1, 1, 1, 1), AMP=2e-4, BOUNCE=1e-3,
                    ls_init_value=1e-5, enable_CB=False, input_channels=20, dim_head=32):
    return to_attr({
        "name": "SASTRNN", "compile": None, "input_channels": input_channels, "enable_masking": False,
        "partition_split_32": 2, "embed_dim": embed_dim, "dim_multiplier": [1, 2, 4, 8], "num_blocks": list(num_blocks),
        "T_max_chrono_init": [4, 8, 16, 32], "stem": {"patch_size": 4}, "in_res_hw": tuple(in_res_hw),
        "stage": {
            "downsample": {"type": "patch", "overlap": True, "norm_affine": True},
            "attention": {"use_torch_mha": False, "partition_size": tuple(partition_size), "dim_head": dim_head,
                          "attention_bias": True, "mlp_activation": "gelu", "mlp_gated": False, "mlp_bias": True,
                          "mlp_ratio": 4, "drop_mlp": 0, "drop_path": 0, "ls_init_value": ls_init_value,
                          "enable_CB": enable_CB, "AMP": AMP, "BOUNCE": BOUNCE},
            "lstm": {"dws_conv": False, "dws_conv_only_hidden": True, "dws_conv_kernel_size": 3, "drop_cell_update": 0},
        },
    })
