"""Host-side mirror of models/layers/SAST/SAST.py for the MI355X path.

Same class names, constructor arguments, forward signatures and state_dict keys as the reference
(SURVEY.md §8b, App. D-10); the arithmetic runs in the HIP kernels of libsast_hip.so:

  SAST_block.forward  (SAST.py:98-164) ->  sast_add_rows -> sast_score_stp -> sast_select (window)
                                           -> sast_mswsa (window) -> sast_select (grid) -> sast_mswsa (grid)

Nothing is ever window/grid-partitioned in memory: tokens stay in image layout (B,H,W,C) and the
kernels address them through the partition map, so the reference's five permute+contiguous round
trips per block (ops.py:189-220) disappear.  Selection results stay on the device
(functional.Selection); `index_count` is returned as a 0-dim device tensor unless
`sync_index_count=True` (then a python int like the reference, at the price of a host sync).
"""
from __future__ import annotations

import math
from typing import List, Optional, Tuple

import torch
from torch import nn

from .. import functional as SF
from .ops import LayerNorm, LayerScale, MLP, cfg_get


class PositiveLinear(nn.Module):
    """SAST.py:305-328: linear layer whose effective weights are exp(weight).  Inside SAST_block it is
    evaluated by the fused scoring kernel (csrc/k_rows.hip:controls_fwd_kernel)."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_features))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            bound = 1 / math.sqrt(self.in_features)
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, input):
        """SAST.py:325-328: F.linear(input, exp(weight), bias).  Stand-alone use only (a (B,20) x (20,C) product, host-side
        torch ops on whatever device the module lives on); inside SAST_block the same expression is evaluated by the fused
        scoring launch (csrc/common.cuh:controls_fwd_elem), which this method is never called from."""
        return nn.functional.linear(input, self.weight.exp(), self.bias)


def get_score_index_2d21d(x: torch.Tensor, d: float, b: float) -> torch.Tensor:
    """SAST.py:258-267 (host utility kept for API compatibility; the hot path uses sast_select)."""
    nz = torch.nonzero(x >= d / (1 + b))
    if x.shape[0] == 1:
        return nz[:, 1]
    return nz[:, 0] * x.shape[-1] + nz[:, 1]


def get_score_index_with_padding(x: torch.Tensor, d: float, b: float):
    """SAST.py:270-281 (host utility kept for API compatibility)."""
    gt = x >= d / (1 + b)
    K = torch.sum(gt, dim=1)
    top = torch.topk(x, k=int(K.max()), dim=1, largest=True, sorted=False)[1]
    base = torch.arange(0, x.shape[0] * x.shape[1], x.shape[1], device=x.device).view(-1, 1)
    nz = torch.nonzero(gt)
    return (top + base).view(-1), nz[:, 0] * x.shape[-1] + nz[:, 1], K


class DropPath(nn.Module):
    """layers/drop.py:157-169 (stochastic depth): in MS_WSA it acts on (kept rows, C) tensors, i.e. per kept TOKEN.  No parameters; the
    scaling lives in the kernels (include/sast_hip.h: SastMswsaArgs.drop1 / drop2), this module only draws the row factors."""

    def __init__(self, drop_prob: float = 0., scale_by_keep: bool = True):
        super().__init__()
        self.drop_prob, self.scale_by_keep = drop_prob, scale_by_keep

    def row_factors(self, rows: int, device) -> torch.Tensor:
        keep = 1.0 - self.drop_prob
        f = torch.empty(rows, device=device).bernoulli_(keep)
        return f.div_(keep) if (keep > 0.0 and self.scale_by_keep) else f

    def extra_repr(self):
        return f"p={self.drop_prob:g}, scale_by_keep={self.scale_by_keep}"


def _glu_activation_name(act) -> str:
    """the reference hands MS_WSA an activation CLASS (get_act_layer(name), SAST.py:55); here a name, such a class, or None (= gelu)"""
    if act is None:
        return "gelu"
    if isinstance(act, str):
        name = act
    else:
        name = {nn.GELU: "gelu", nn.ReLU: "relu", nn.SiLU: "silu", nn.Sigmoid: "sigmoid", nn.Tanh: "tanh", nn.Mish: "mish", nn.ReLU6: "relu6",
                nn.LeakyReLU: "leaky_relu", nn.ELU: "elu", nn.CELU: "celu", nn.SELU: "selu", nn.Hardsigmoid: "hard_sigmoid",
                nn.Hardswish: "hard_swish", nn.PReLU: "prelu"}.get(act)
        if name is None and isinstance(act, type) and issubclass(act, nn.PReLU):       # the reference's own subclass (activations.py:124)
            name = "prelu"
    if name not in SF.GLU_ACTIVATIONS:
        raise NotImplementedError(f"sast_amd: MLP activation {act!r}: the GLU epilogues implement {sorted(SF.GLU_ACTIVATIONS)}")
    return name


class MS_WSA(nn.Module):
    """Masked Sparse Window multi-head Self-Attention, channels-last (SAST.py:167-255)."""

    def __init__(self, dim: int, dim_head: int = 32, bias: bool = True, sub_layer_params=None, norms=None):
        super().__init__()
        if dim_head % 4 or not 4 <= dim_head <= 32 or dim % dim_head:
            # SAST.py:35,171-179 accept any divisor of dim; the attention kernels hold a head's q / k / v in one 32-wide plane row
            # (narrower heads are zero-padded), so heads of 4, 8, ... 32 channels run -- the shipped models use 32 (tiny / base / large: 24)
            raise NotImplementedError(f"sast_amd: dim_head {dim_head} (dim {dim}): the attention kernels take head widths that are multiples "
                                      "of 4 up to 32 and divide dim")
        self.num_heads = dim // dim_head
        self.dim_head = dim_head
        self.scale = dim_head ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=bias)
        self.proj = nn.Linear(dim, dim, bias=bias)
        self.norm1 = norms[0]
        ls_init_value, drop_path, mlp_expand_ratio, mlp_act_layer, mlp_bias, drop_mlp = sub_layer_params
        if not 0.0 <= drop_path < 1.0:
            raise ValueError(f"drop_path must be in [0, 1), got {drop_path}")
        self.mlp_activation = _glu_activation_name(mlp_act_layer)
        self.ls1 = LayerScale(dim=dim, init_values=ls_init_value) if ls_init_value > 0 else nn.Identity()
        self.drop1 = DropPath(drop_path) if drop_path > 0 else nn.Identity()
        self.norm2 = norms[1]
        self.mlp = MLP(dim=dim, channel_last=True, expansion_ratio=mlp_expand_ratio, act_layer=self.mlp_activation, bias=mlp_bias,
                       drop_prob=drop_mlp)
        self.ls2 = LayerScale(dim=dim, init_values=ls_init_value) if ls_init_value > 0 else nn.Identity()
        self.drop2 = DropPath(drop_path) if drop_path > 0 else nn.Identity()
        self.drop_path_override = None     # tests: fixed (d1, d2, mlp_mask) instead of a fresh draw (see _drop_path_factors)
        # aliased container, same state_dict duplicates as the reference (SAST.py:194)
        self.sub_layers = nn.ModuleList([self.ls1, self.drop1, self.norm2, self.mlp, self.ls2, self.drop2])
        self.eps = 1e-6
        # attention_bias: False / mlp_bias: False (SAST.py:180-181, ops.py:128,160-166): the kernels always add a bias vector, a linear
        # without one gets a resident zero vector (not a parameter, not in the state_dict; its "gradient" goes to a throw-away buffer)
        inner = self.mlp.inner_dim
        for name, lin, n in (("_zero_qkv_b", self.qkv, 3 * dim), ("_zero_proj_b", self.proj, dim),
                             ("_zero_fc1_b", self.mlp.net[0].proj, 2 * inner), ("_zero_fc2_b", self.mlp.net[2], dim)):
            if lin.bias is None:
                self.register_buffer(name, torch.zeros(n), persistent=False)

    def kernel_params(self) -> dict:
        return dict(ln1_w=self.norm1.weight, ln1_b=self.norm1.bias, ln2_w=self.norm2.weight, ln2_b=self.norm2.bias,
                    qkv_w=self.qkv.weight, qkv_b=self.qkv.bias if self.qkv.bias is not None else self._zero_qkv_b,
                    proj_w=self.proj.weight, proj_b=self.proj.bias if self.proj.bias is not None else self._zero_proj_b,
                    ls1=getattr(self.ls1, "gamma", None), fc1_w=self.mlp.net[0].proj.weight,
                    fc1_b=self.mlp.net[0].proj.bias if self.mlp.net[0].proj.bias is not None else self._zero_fc1_b,
                    fc2_w=self.mlp.net[2].weight, fc2_b=self.mlp.net[2].bias if self.mlp.net[2].bias is not None else self._zero_fc2_b,
                    ls2=getattr(self.ls2, "gamma", None), act_w=getattr(getattr(self.mlp.net[0], "act_layer", None), "weight", None))

    def _drop_path_factors(self, rows: int, device):
        """training-mode randomness of the layer, in the reference's order (SAST.py:232-248): DropPath on the attention branch, nn.Dropout on
        the MLP hidden (`drop_mlp`, ops.py:167), DropPath on the MLP branch.  DropPath (layers/drop.py): one Bernoulli(keep_prob) draw per
        KEPT ROW, divided by keep_prob; the number of kept rows lives on the device, so the draws cover the row upper bound and entry m
        serves the m-th kept row.  -> None, or (d1, d2, mlp_mask) with None for what is off."""
        if not self.training:
            return None
        if self.drop_path_override is not None:
            return self.drop_path_override
        dp, pm = isinstance(self.drop1, DropPath), self.mlp.net[1].p
        if not dp and pm == 0.0:
            return None
        d1 = self.drop1.row_factors(rows, device) if dp else None
        mask = torch.empty(rows, self.mlp.inner_dim, device=device).bernoulli_(1.0 - pm).div_(1.0 - pm) if pm > 0.0 else None
        d2 = self.drop2.row_factors(rows, device) if dp else None
        return d1, d2, mask

    def forward_image(self, x: torch.Tensor, sel: SF.Selection, enable_CB: bool = False, fused: bool = True) -> torch.Tensor:
        """device path: x (B,H,W,C) in IMAGE layout + device-side selection."""
        return SF.mswsa(x, sel, self.norm1.eps, self.kernel_params(), x.shape[1] * x.shape[2] if enable_CB else 0, self.dim_head, fused,
                        self.mlp_activation, self._drop_path_factors(x.numel() // x.shape[-1], x.device))

    def forward(self, x: torch.Tensor, index_window: torch.Tensor, index_token: torch.Tensor, padding_index: torch.Tensor,
                asy_index: torch.Tensor, M: int, B, enable_CB: bool) -> torch.Tensor:
        """reference signature (SAST.py:199-201): x (B*N, T, C) already partitioned, reference index lists.
        The top-k fillers (index_token / padding_index) are semantically inert and ignored."""
        shape = x.shape
        N, C = x.shape[0], x.shape[-1]
        x3 = x.reshape(N, -1, C)
        T = x3.shape[1]
        K = torch.bincount(torch.div(asy_index, T, rounding_mode='floor'), minlength=len(index_window)) if len(index_window) \
            else torch.zeros(0, dtype=torch.long, device=x.device)
        sel = SF.selection_from_index_lists(index_window, asy_index, K, N, T, x.device)
        # Context Broadcasting averages over the tokens of one sample = N*T/B consecutive partitioned tokens (SAST.py:244-245)
        out = SF.mswsa(x3.reshape(1, N * T, 1, C), sel, self.norm1.eps, self.kernel_params(), N * T // int(B) if enable_CB else 0, self.dim_head,
                       mlp_activation=self.mlp_activation, drop_path=self._drop_path_factors(N * T, x.device))
        return out.view(*shape)


FUSED_FORWARD_MAX_AMP = 5e-3


class SAST_block(nn.Module):
    """SAST block = two SAST layers (window, then grid) sharing one scoring module (SAST.py:24-164)."""

    def __init__(self, dim: int, attention_cfg, first_block: bool = False, sync_index_count: bool = False):
        super().__init__()
        norm_eps = cfg_get(attention_cfg, 'norm_eps', 1e-5)
        partition_size = cfg_get(attention_cfg, 'partition_size', required=True)
        dim_head = cfg_get(attention_cfg, 'dim_head', 32)
        attention_bias = cfg_get(attention_cfg, 'attention_bias', True)
        mlp_act_string = cfg_get(attention_cfg, 'mlp_activation', required=True)
        mlp_bias = cfg_get(attention_cfg, 'mlp_bias', True)
        mlp_expand_ratio = cfg_get(attention_cfg, 'mlp_ratio', 4)
        drop_path = cfg_get(attention_cfg, 'drop_path', 0.0)
        drop_mlp = cfg_get(attention_cfg, 'drop_mlp', 0.0)
        ls_init_value = cfg_get(attention_cfg, 'ls_init_value', 1e-5)
        if mlp_act_string not in SF.GLU_ACTIVATIONS:       # layers/create_act.py:62-79 knows 17 names; the GLU epilogues implement these
            raise NotImplementedError(f"sast_amd: mlp_activation {mlp_act_string!r}: the GLU epilogues implement {sorted(SF.GLU_ACTIVATIONS)}")
        if isinstance(partition_size, int):
            partition_size = (partition_size, partition_size)
        else:
            partition_size = tuple(partition_size)
            assert len(partition_size) == 2
        self.partition_size = partition_size
        SF.mask_words(partition_size[0] * partition_size[1])      # raises beyond 256 tokens (gen4 with partition_split_32 1 has 240)
        sub_layer_params = (ls_init_value, drop_path, mlp_expand_ratio, mlp_act_string, mlp_bias, drop_mlp)
        self.enable_CB = cfg_get(attention_cfg, 'enable_CB', False)
        mk_norm = lambda: LayerNorm(dim, eps=norm_eps)
        self.win_attn = MS_WSA(dim, dim_head=dim_head, bias=attention_bias, sub_layer_params=sub_layer_params,
                               norms=[mk_norm(), mk_norm()])
        self.grid_attn = MS_WSA(dim, dim_head=dim_head, bias=attention_bias, sub_layer_params=sub_layer_params,
                                norms=[mk_norm(), mk_norm()])
        if first_block:
            self.to_scores = nn.Linear(dim, dim)
            self.to_controls = PositiveLinear(20, dim, bias=False)
            torch.nn.init.constant_(self.to_controls.weight, 1)
            self.act = nn.ReLU()
        self.amp_value = cfg_get(attention_cfg, 'AMP', 2e-4)
        # the one-kernel forward of the MS-WSA layers pays when most tokens survive the selection; the kept fraction is set by AMP
        # (2e-4 -> ~100 %, 2e-3 -> ~58 %, 2e-2 -> ~29 %).  Measured crossover between 2e-3 and 2e-2 (profiles/r04_j).
        self.fused_forward = self.amp_value <= FUSED_FORWARD_MAX_AMP
        self.bounce_value = cfg_get(attention_cfg, 'BOUNCE', 1e-3)
        self.first_block = first_block
        self.sync_index_count = sync_index_count
        self.B, self.N, self.dim = None, None, dim

    # -- fused entry used by the backbone: `xp` already holds x + pos_emb (added by the LayerNorm kernel)
    def forward_posadded(self, xp: torch.Tensor, r: torch.Tensor, index_list):
        B, H, W, C = xp.shape
        ph, pw = self.partition_size
        self.B, self.N = B, H * W // (ph * pw)
        if self.first_block:
            xw, tok = SF.score_stp(xp, r, self.to_scores.weight, self.to_scores.bias, self.to_controls.weight, self.amp_value)
            sel1, sel2 = SF.select_pair(tok, B, H, W, ph, pw, self.bounce_value)
        else:
            xw = xp
            sel1, sel2 = index_list
            if not isinstance(sel1, SF.Selection):
                raise TypeError("sast_amd: index_list must be the [Selection, Selection] pair returned by the first block")
        x = self.win_attn.forward_image(xw, sel1, self.enable_CB, self.fused_forward)
        x = self.grid_attn.forward_image(x, sel2, self.enable_CB, self.fused_forward)
        count = SF.DeviceCount((sel1.counts[2], sel2.counts[2]))   # SAST.py:136,159 (floor per layer), summed lazily on the device
        if self.sync_index_count:
            count = count.item()
        return x, count, [sel1, sel2]

    def forward(self, x: torch.Tensor, pos_emb, r: torch.Tensor, index_list) -> Tuple[torch.Tensor, object, List]:
        """x (B,H,W,C) NHWC, pos_emb: module returning the (B,H,W,C) table (sast_rnn.py:215-219) or a tensor,
        r (B,20).  -> (x, index_count, [list1, list2])"""
        table = pos_emb.table_for(x) if hasattr(pos_emb, "table_for") else pos_emb(x)[0]
        xp = SF.add_pos_embedding(x, table.contiguous())
        return self.forward_posadded(xp, r, index_list)
