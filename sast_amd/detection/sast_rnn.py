"""Host-side mirror of models/detection/recurrent_backbone/sast_rnn.py (RNNDetector and friends).

Same constructor configs, forward signatures and state_dict keys; all arithmetic is in libsast_hip.so.
Feature maps returned to the caller are logical NCHW tensors in channels-last memory (zero-copy views
of the NHWC buffers the kernels work on).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from .. import functional as SF
from ..layers.ops import cfg_get, get_downsample_layer_Cf2Cl
from ..layers.rnn import DWSConvLSTM2d
from ..layers.sast import SAST_block


def non_zero_ratio(x: torch.Tensor) -> torch.Tensor:
    """sast_rnn.py:45-60 -> (B,4,20)"""
    return SF.non_zero_ratio(x)


class PositionEmbeddingSine(nn.Module):
    """sast_rnn.py:180-219.  The constant table is built once on the host (it is not a parameter or a
    buffer in the reference either, :192) and kept on the device as an (H*W, C) row table."""

    def __init__(self, num_pos_feats=64, temperature=10000, normalize=False, scale=None, input_size=(128, 128, 128)):
        super().__init__()
        if scale is not None and normalize is False:
            raise ValueError("normalize should be True if scale is passed")
        self.num_pos_feats, self.temperature, self.normalize = num_pos_feats, temperature, normalize
        self.scale = 2 * math.pi if scale is None else scale
        self.pos_embedding = self.generate_position_embedding(input_size)
        self._tables = {}

    def generate_position_embedding(self, input_size):
        _, H, W = input_size
        yy = torch.arange(1, H + 1, dtype=torch.float32).view(1, H, 1).expand(1, H, W)
        xx = torch.arange(1, W + 1, dtype=torch.float32).view(1, 1, W).expand(1, H, W)
        if self.normalize:
            eps = 1e-6
            yy = (yy - 0.5) / (yy[:, -1:, :] + eps) * self.scale
            xx = (xx - 0.5) / (xx[:, :, -1:] + eps) * self.scale
        k = torch.arange(self.num_pos_feats, dtype=torch.float32)
        div = self.temperature ** (2 * (k // 2) / self.num_pos_feats)
        px, py = xx[..., None] / div, yy[..., None] / div
        px = torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), dim=4).flatten(3)
        py = torch.stack((py[..., 0::2].sin(), py[..., 1::2].cos()), dim=4).flatten(3)
        return torch.cat((py, px), dim=3)

    def table_for(self, x: torch.Tensor) -> torch.Tensor:
        """(H*W, C) device table for a (B,H,W,C) input (slice [:H,:W] of the full table, sast_rnn.py:218)."""
        H, W = x.shape[1:3]
        key = (H, W, x.device)
        t = self._tables.get(key)
        if t is None:
            t = self.pos_embedding[0, :H, :W, :].contiguous().view(H * W, -1).to(x.device)
            self._tables[key] = t
        return t

    def forward(self, x):
        B, H, W = x.shape[:3]
        return self.table_for(x).view(1, H, W, -1).expand(B, -1, -1, -1)


class SASTAttentionPairCl(nn.Module):
    """sast_rnn.py:164-178"""

    def __init__(self, dim: int, skip_first_norm: bool, attention_cfg, first_block: bool = False):
        super().__init__()
        self.att = SAST_block(dim=dim, attention_cfg=attention_cfg, first_block=first_block)
        self.first_block = first_block

    def forward(self, x, pos_emb, r, index_list):
        x, p_loss, index_list = self.att(x, pos_emb, r, index_list)
        return x, p_loss, r, index_list


class RNNDetectorStage(nn.Module):
    """sast_rnn.py:221-287.  NCHW in / out."""

    def __init__(self, dim_in: int, stage_dim: int, spatial_downsample_factor: int, num_blocks: int,
                 enable_token_masking: bool, T_max_chrono_init: Optional[int], stage_cfg, overload_size, enable_lstm: bool):
        super().__init__()
        assert isinstance(num_blocks, int) and num_blocks > 0
        downsample_cfg, lstm_cfg, attention_cfg = stage_cfg.downsample, stage_cfg.lstm, stage_cfg.attention
        self.downsample_cf2cl = get_downsample_layer_Cf2Cl(dim_in=dim_in, dim_out=stage_dim,
                                                           downsample_factor=spatial_downsample_factor,
                                                           downsample_cfg=downsample_cfg)
        self.att_blocks = nn.ModuleList([
            SASTAttentionPairCl(dim=stage_dim, skip_first_norm=(i == 0), attention_cfg=attention_cfg, first_block=(i == 0))
            for i in range(num_blocks)])
        self.lstm = DWSConvLSTM2d(dim=stage_dim, dws_conv=lstm_cfg.dws_conv, dws_conv_only_hidden=lstm_cfg.dws_conv_only_hidden,
                                  dws_conv_kernel_size=lstm_cfg.dws_conv_kernel_size,
                                  cell_update_dropout=cfg_get(lstm_cfg, 'drop_cell_update', 0)) if enable_lstm else None
        self.pos_emb = PositionEmbeddingSine(stage_dim // 2, normalize=True, input_size=overload_size)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, 1, stage_dim), requires_grad=True) if enable_token_masking else None
        self.last_index_list = None
        if self.mask_token is not None:
            torch.nn.init.normal_(self.mask_token, std=.02)

    def forward_nhwc(self, x_nhwc, h_and_c_previous, r, token_mask=None):
        """x (B,H,W,Cin) fp32 channels-last -> (h (B,H',W',C), (h,c), P)"""
        ds = self.downsample_cf2cl
        H, W = x_nhwc.shape[1] // ds.factor, x_nhwc.shape[2] // ds.factor
        probe = x_nhwc.new_empty((1, H, W, 0))
        table = self.pos_emb.table_for(probe)
        x = ds.forward_nhwc(x_nhwc, table)                       # LN(conv(x)) + pos_emb of the first block
        if token_mask is not None:                               # sast_rnn.py:271-273 (the rows already carry pos_emb)
            assert self.mask_token is not None, 'No mask token present in this stage'
            x = SF.mask_token(x, token_mask, self.mask_token, table)
        P = 0
        index_list = None
        for i, blk in enumerate(self.att_blocks):
            if i > 0:
                x = SF.add_pos_embedding(x, table)                # every block adds the table (SAST.py:105)
            x, p_loss, index_list = blk.att.forward_posadded(x, r, index_list)
            P = P + p_loss
        self.last_index_list = index_list        # the stage's [Selection, Selection] (device masks; parity tests read them)
        if self.lstm is not None:
            # h1 goes to the next stage and, through the state, to the FPN / the next time step: two aliases, one per consumer
            h1, h1b, c1 = self.lstm.forward_nhwc(x, h_and_c_previous, two_h=True)
            return h1, (h1b, c1), P
        return x, (x, x), P

    def forward(self, x: torch.Tensor, h_and_c_previous=None, token_mask: Optional[torch.Tensor] = None, r: torch.Tensor = None):
        x_nhwc = SF.as_nhwc(x) if x.dtype == torch.float32 else SF.nchw_to_nhwc_float(x)
        hc = None
        if h_and_c_previous is not None:
            hc = (SF.as_nhwc(h_and_c_previous[0]), SF.as_nhwc(h_and_c_previous[1]))
        h, (h1, c1), P = self.forward_nhwc(x_nhwc, hc, r, token_mask)
        return SF.as_nchw_view(h), (SF.as_nchw_view(h1), SF.as_nchw_view(c1)), P


class RNNDetector(nn.Module):
    """sast_rnn.py:67-162"""

    def __init__(self, mdl_config):
        super().__init__()
        in_channels = mdl_config.input_channels
        embed_dim = mdl_config.embed_dim
        dim_multiplier_per_stage = tuple(mdl_config.dim_multiplier)
        num_blocks_per_stage = tuple(mdl_config.num_blocks)
        T_max_chrono_init_per_stage = tuple(mdl_config.T_max_chrono_init)
        enable_masking = mdl_config.enable_masking
        num_stages = len(num_blocks_per_stage)
        assert num_stages == 4 and isinstance(embed_dim, int)
        assert num_stages == len(dim_multiplier_per_stage) == len(T_max_chrono_init_per_stage)
        compile_cfg = cfg_get(mdl_config, 'compile', None)
        if compile_cfg is not None and cfg_get(compile_cfg, 'enable', False):
            # compile.enable asks the reference for torch.compile (sast_rnn.py:86-95 / yolo_pafpn.py:40-45): an optimisation, not a change
            # of results.  There is no tracing compiler here -- the forward is hand-written launches and the step replays as hipGraphs
            import warnings
            warnings.warn("sast_amd: compile.enable is ignored (no torch.compile: capture the step in a hipGraph, sast_amd.training.TrainStep)")
        input_dim = in_channels
        patch_size = mdl_config.stem.patch_size
        stride = 1
        self.stage_dims = [embed_dim * x for x in dim_multiplier_per_stage]
        self.stages = nn.ModuleList()
        self.strides = []
        in_res_h, in_res_w = mdl_config.in_res_hw
        self.in_res_hw = (int(in_res_h), int(in_res_w))
        size = (1, in_res_h, in_res_w)
        for stage_idx, (num_blocks, T_max) in enumerate(zip(num_blocks_per_stage, T_max_chrono_init_per_stage)):
            f = patch_size if stage_idx == 0 else 2
            stage_dim = self.stage_dims[stage_idx]
            size = (1, size[1] // f, size[2] // f)
            self.stages.append(RNNDetectorStage(dim_in=input_dim, stage_dim=stage_dim, spatial_downsample_factor=f,
                                                num_blocks=num_blocks, enable_token_masking=enable_masking and stage_idx == 0,
                                                T_max_chrono_init=T_max, stage_cfg=mdl_config.stage, overload_size=size,
                                                enable_lstm=True))
            stride = stride * f
            self.strides.append(stride)
            input_dim = stage_dim
        self.num_stages = num_stages

    def get_stage_dims(self, stages: Tuple[int, ...]) -> Tuple[int, ...]:
        idx = [x - 1 for x in stages]
        assert min(idx) >= 0 and max(idx) < len(self.stages), idx
        return tuple(self.stage_dims[i] for i in idx)

    def get_strides(self, stages: Tuple[int, ...]) -> Tuple[int, ...]:
        idx = [x - 1 for x in stages]
        assert min(idx) >= 0 and max(idx) < len(self.stages), idx
        return tuple(self.strides[i] for i in idx)

    def forward_nhwc(self, x: torch.Tensor, prev_states=None, token_mask=None, cut_before_stage=None):
        """fused path: returns NHWC feature maps {stage: (B,H,W,C)}, states [(h,c)] NHWC, P list.
        cut_before_stage (training.TrainStep; an int or several): the input of that stage (0-based) enters it as a detached leaf, so the
        backward pass can run in segments around it; the pairs (upstream tensor, leaf) are left in `self.last_cuts` {stage: pair}
        (`self.last_cut`: the pair of the LAST such stage)."""
        self.last_cut, self.last_cuts = None, {}
        cuts = () if cut_before_stage is None else ((cut_before_stage,) if isinstance(cut_before_stage, int) else tuple(cut_before_stage))
        if prev_states is None:
            prev_states = [None] * self.num_stages
        assert len(prev_states) == self.num_stages
        # an event tensor smaller than in_res_hw stands for its zero padding (the reference pads it first,
        # modules/detection.py:143-144 / utils/padding.py:29-53): ratios, cast, layout change and padding in two passes over x
        pad = self.in_res_hw if (x.shape[-2] < self.in_res_hw[0] or x.shape[-1] < self.in_res_hw[1]) else None
        if not hasattr(self, "_prep_ws"):
            self._prep_ws = {}                # scratch of the input kernel, owned by this module (one per device and batch size)
        # ratios, cast, padding and layout change in one launch that reads x once; a uint8 tensor (the dataset's storage type) stays bytes:
        # the stem conv's loaders widen them, the fp32 copy of the input is never written
        r, xin = SF.input_prep(x, pad, self._prep_ws, keep_bytes=SF.STEM_U8)
        states, output, P = [], {}, []
        for i, stage in enumerate(self.stages):
            if i in cuts and torch.is_grad_enabled() and xin.requires_grad:
                leaf = xin.detach().requires_grad_(True)
                self.last_cut = self.last_cuts[i] = (xin, leaf)
                xin = leaf
            xin, state, p = stage.forward_nhwc(xin, prev_states[i], r[:, i], token_mask if i == 0 else None)   # sast_rnn.py:157
            states.append(state)
            output[i + 1] = state[0]
            P.append(p)
        return output, states, P

    def forward(self, x: torch.Tensor, prev_states=None, token_mask: Optional[torch.Tensor] = None):
        """x (B,20,H,W) NCHW any of {uint8,int32,float32}; -> ({1..4: h NCHW}, [(h,c)], P)  (sast_rnn.py:144-162)"""
        ps = None
        if prev_states is not None:
            ps = [None if s is None else (SF.as_nhwc(s[0]), SF.as_nhwc(s[1])) for s in prev_states]
        out, states, P = self.forward_nhwc(x, ps, token_mask)
        out = {k: SF.as_nchw_view(v) for k, v in out.items()}
        states = [(SF.as_nchw_view(h), SF.as_nchw_view(c)) for h, c in states]
        return out, states, P
