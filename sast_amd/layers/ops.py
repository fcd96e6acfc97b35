"""Host-side mirror of the reference's models/layers/SAST/ops.py operator surface.

The partition functions are pure index maps (views + one copy) kept for API compatibility and
tests; the fused HIP path never materialises a partitioned tensor (kernels address tokens through
the window/grid map, see csrc/common.cuh PartMap).  MLP / GLU / LayerScale / LayerNorm are
parameter containers with the reference's state_dict names; their arithmetic runs inside the
fused MS-WSA kernels (sast_amd/functional.py:mswsa).
"""
from __future__ import annotations

import math
from typing import Tuple

import torch
from torch import nn

from .. import functional as SF


def cfg_get(cfg, key, default=None, required=False):
    """attention_cfg / downsample_cfg may be an omegaconf DictConfig, a dict or any attribute bag."""
    if cfg is None:
        val = None
    elif isinstance(cfg, dict):
        val = cfg.get(key, None)
    elif hasattr(cfg, "get"):
        val = cfg.get(key, None)
    else:
        val = getattr(cfg, key, None)
    if val is None:
        if required:
            raise KeyError(f"missing required config key '{key}'")
        return default
    return val


def nChw_2_nhwC(x: torch.Tensor) -> torch.Tensor:
    """ops.py:19-24"""
    assert x.ndim == 4
    return SF.as_nhwc(x)


def nhwC_2_nChw(x: torch.Tensor) -> torch.Tensor:
    """ops.py:27-30 -- returns the NCHW *view* of the channels-last buffer (same values, no copy)."""
    assert x.ndim == 4
    return SF.as_nchw_view(x)


def window_partition(x: torch.Tensor, window_size: Tuple[int, int]) -> torch.Tensor:
    """ops.py:189-195"""
    B, H, W, C = x.shape
    h, w = window_size
    assert H % h == 0, f'height ({H}) must be divisible by window ({h})'
    assert W % w == 0, f'width ({W}) must be divisible by window ({w})'
    return x.reshape(B, H // h, h, W // w, w, C).transpose(2, 3).reshape(-1, h, w, C)


def window_reverse(windows: torch.Tensor, window_size: Tuple[int, int], img_size: Tuple[int, int]) -> torch.Tensor:
    """ops.py:198-203"""
    H, W = img_size
    h, w = window_size
    C = windows.shape[-1]
    return windows.reshape(-1, H // h, W // w, h, w, C).transpose(2, 3).reshape(-1, H, W, C)


def grid_partition(x: torch.Tensor, grid_size: Tuple[int, int]) -> torch.Tensor:
    """ops.py:206-212"""
    B, H, W, C = x.shape
    h, w = grid_size
    assert H % h == 0, f'height {H} must be divisible by grid {h}'
    assert W % w == 0, f'width {W} must be divisible by grid {w}'
    return x.reshape(B, h, H // h, w, W // w, C).permute(0, 2, 4, 1, 3, 5).reshape(-1, h, w, C)


def grid_reverse(windows: torch.Tensor, grid_size: Tuple[int, int], img_size: Tuple[int, int]) -> torch.Tensor:
    """ops.py:215-220"""
    H, W = img_size
    h, w = grid_size
    C = windows.shape[-1]
    return windows.reshape(-1, H // h, W // w, h, w, C).permute(0, 3, 1, 4, 2, 5).reshape(-1, H, W, C)


class _FusedOnly(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError(f"{type(self).__name__} is a parameter container in sast_amd: its arithmetic is fused into the "
                           "MS-WSA HIP kernels; call MS_WSA / SAST_block instead")


class LayerNorm(nn.LayerNorm):
    """layers/norm.py:44-56 (plain F.layer_norm over channels).  Container; fused in the HIP kernels."""

    def __init__(self, num_channels, eps=1e-6, affine=True):
        super().__init__(num_channels, eps=eps, elementwise_affine=affine)

    def forward(self, x):
        raise RuntimeError("LayerNorm is fused into the sast_amd HIP kernels; it is not callable on its own")


class LayerScale(_FusedOnly):
    """ops.py:178-186"""

    def __init__(self, dim: int, init_values: float = 1e-5, inplace: bool = False):
        super().__init__()
        self.inplace = inplace
        self.gamma = nn.Parameter(init_values * torch.ones(dim))


class PReLUSlope(_FusedOnly):
    """layers/activations.py:124-131 (nn.PReLU, num_parameters 1, init 0.25): the parameter holder of `mlp_activation: prelu` -- same name
    (`act_layer.weight`) and shape as the reference's module, applied inside the GLU epilogues"""

    def __init__(self, num_parameters: int = 1, init: float = 0.25, inplace: bool = False):
        super().__init__()
        assert num_parameters == 1
        self.weight = nn.Parameter(torch.full((1,), float(init)))


class GLU(_FusedOnly):
    """ops.py:111-137 (channel-last only): proj = Linear(dim_in, 2*dim_out); out = value * act(gate)."""

    def __init__(self, dim_in: int, dim_out: int, channel_last: bool = True, act_layer=None, bias: bool = True):
        super().__init__()
        assert channel_last, "sast_amd implements the channels-last MLP of MS_WSA only"
        self.proj = nn.Linear(dim_in, dim_out * 2, bias=bias)
        if act_layer == "prelu" or act_layer is nn.PReLU or (isinstance(act_layer, type) and issubclass(act_layer, nn.PReLU)):
            self.act_layer = PReLUSlope()      # (the parameter-free activations add nothing to the state_dict)


class MLP(_FusedOnly):
    """ops.py:140-175 with gated=True (the only form SAST_block builds, SAST.py:190): GLU-GELU MLP,
    inner = floor(dim*ratio*2/3/32)*32."""

    def __init__(self, dim: int, channel_last: bool = True, expansion_ratio: int = 4, act_layer=None, gated: bool = True,
                 bias: bool = True, drop_prob: float = 0.):
        super().__init__()
        assert gated and channel_last, "sast_amd implements the gated channels-last MLP of MS_WSA only"
        if not 0.0 <= drop_prob < 1.0:
            raise ValueError(f"drop_mlp must be in [0, 1), got {drop_prob}")
        inner = math.floor(int(dim * expansion_ratio) * 2 / 3 / 32) * 32
        self.inner_dim = inner
        self.net = nn.Sequential(GLU(dim, inner, True, act_layer, bias), nn.Dropout(p=drop_prob), nn.Linear(inner, dim, bias=bias))


class DownsampleBase(nn.Module):
    @staticmethod
    def output_is_normed():
        raise NotImplementedError


def channels_last_conv_weight(cout, cin, k):
    """Parameter of logical shape [Cout,Cin,k,k] stored as [Cout][k][k][Cin] (what the implicit-GEMM loaders read)."""
    w = torch.empty(cout, k, k, cin)
    nn.init.kaiming_uniform_(w.permute(0, 3, 1, 2), a=math.sqrt(5))
    return nn.Parameter(w.permute(0, 3, 1, 2))


class ConvDownsampling_Cf2Cl(DownsampleBase):
    """ops.py:54-95.  NCHW in -> NHWC out: conv(k=2f-1, stride f, replicate pad f-1, no bias) + LayerNorm(eps 1e-5).
    downsample_cfg.overlap False: k = f without padding; norm_affine False: LayerNorm without weight / bias (neither is in a shipped YAML).

    `forward_nhwc(x_nhwc, pe)` is the fused entry used by the backbone: it also adds the block's
    position table (SAST.py:105) in the LayerNorm kernel.
    """

    def __init__(self, dim_in: int, dim_out: int, downsample_factor: int, downsample_cfg=None):
        super().__init__()
        assert isinstance(dim_out, int) and isinstance(dim_in, int)
        assert downsample_factor in (2, 4, 8)
        norm_affine = cfg_get(downsample_cfg, 'norm_affine', True)
        overlap = cfg_get(downsample_cfg, 'overlap', True)
        self.factor = downsample_factor
        k = (downsample_factor - 1) * 2 + 1 if overlap else downsample_factor      # ops.py:70-76 (no overlap: k = f, no padding)
        self.conv = nn.Module()
        self.conv.weight = channels_last_conv_weight(dim_out, dim_in, k)
        self.norm = LayerNorm(num_channels=dim_out, eps=1e-5, affine=norm_affine)
        if not norm_affine:    # the kernels always apply an affine: resident ones / zeros (not parameters, not in the state_dict)
            self.register_buffer("_ln_ones", torch.ones(dim_out), persistent=False)
            self.register_buffer("_ln_zeros", torch.zeros(dim_out), persistent=False)

    def forward_nhwc(self, x_nhwc: torch.Tensor, pe=None) -> torch.Tensor:
        ln_w, ln_b = (self.norm.weight, self.norm.bias) if self.norm.weight is not None else (self._ln_ones, self._ln_zeros)
        return SF.downsample_ln(x_nhwc, self.conv.weight, ln_w, ln_b, pe, self.factor)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x_nhwc = SF.as_nhwc(x) if x.dtype == torch.float32 else SF.nchw_to_nhwc_float(x)
        return self.forward_nhwc(x_nhwc, None)

    @staticmethod
    def output_is_normed():
        return True


def get_downsample_layer_Cf2Cl(dim_in: int, dim_out: int, downsample_factor: int, downsample_cfg) -> DownsampleBase:
    if cfg_get(downsample_cfg, 'type', required=True) == 'patch':
        return ConvDownsampling_Cf2Cl(dim_in=dim_in, dim_out=dim_out, downsample_factor=downsample_factor,
                                      downsample_cfg=downsample_cfg)
    raise NotImplementedError
