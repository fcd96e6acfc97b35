"""ORACLE — test infrastructure only.  NOT part of the product path.

A plain PyTorch-CPU, functional restatement of the SAST hot path (SURVEY.md
§8a rows a1..a13): STP scoring / window+token selection, masked sparse window
self-attention (MS-WSA) in the reference's padded top-k formulation, the
window/grid partitions, GLU-MLP, conv-downsample + LayerNorm, ConvLSTM and the
YOLOX PAFPN.  Every function cites the reference file:line it follows.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module; `sast_amd/` never does (tests/test_no_oracle_in_product.py
enforces it).

Parity pin: the reference ships no tests or golden vectors, so
this oracle is pinned against outputs of the reference itself, imported in the
build container by `tests/golden/make_golden.py`; the captured vectors live in
`tests/golden/*.npz` and `tests/test_oracle_golden.py` checks the oracle against
them (indices exact, floating point `torch.equal` or <=1e-6).

Parameters are passed as a flat dict keyed by the reference's state_dict names
(SURVEY.md App. D-10), so a reference checkpoint can be used unchanged.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]


@dataclass
class AttnCfg:
    """attention_cfg keys read by SAST_block.__init__ (SAST.py:34-44,60,79-80)."""
    partition_size: Tuple[int, int]
    dim_head: int = 32
    norm_eps: float = 1e-5
    amp: float = 2e-4
    bounce: float = 1e-3
    enable_cb: bool = False
    mlp_activation: str = "gelu"     # attention_cfg.mlp_activation (SAST.py:38,55 -> layers/create_act.py:62-79)
    drop_path: float = 0.0           # attention_cfg.drop_path (SAST.py:42,188,193): DropPath on the two residual branches, training mode only
    drop_mlp: float = 0.0            # attention_cfg.drop_mlp (SAST.py:43,191 -> ops.py:167): nn.Dropout on the MLP hidden, training mode only
    training: bool = True
    drop_log: Optional[list] = None    # test plumbing: every DropPath factor vector drawn is appended (call order: layer 1 attention, MLP, layer 2 ...)
    drop_masks: Optional[list] = None  # test plumbing: factor vectors to USE instead of drawing, consumed in the same order


@dataclass
class BackboneCfg:
    """mdl_config keys read by RNNDetector.__init__ (sast_rnn.py:68-130)."""
    in_res_hw: Tuple[int, int]
    partition_size: Tuple[int, int]
    input_channels: int = 20
    embed_dim: int = 64
    dim_multiplier: Tuple[int, ...] = (1, 2, 4, 8)
    num_blocks: Tuple[int, ...] = (1, 1, 1, 1)
    patch_size: int = 4
    attn: AttnCfg = field(init=False)
    amp: float = 2e-4
    bounce: float = 1e-3
    enable_cb: bool = False
    dim_head: int = 32     # 24 in the reference's "small" size (config/experiment/gen1/small.yaml)

    def __post_init__(self):
        self.attn = AttnCfg(partition_size=tuple(self.partition_size), amp=self.amp,
                            bounce=self.bounce, enable_cb=self.enable_cb, dim_head=self.dim_head)

    @property
    def stage_dims(self):
        return [self.embed_dim * m for m in self.dim_multiplier]


# --------------------------------------------------------------------------- a1
def non_zero_ratio(x: Tensor) -> Tensor:
    """sast_rnn.py:45-60.  (B,Cin,H,W) any dtype -> (B,4,Cin) fp32."""
    pooled = []
    cur = F.max_pool2d(x.float(), kernel_size=4, stride=4)
    pooled.append(cur)
    for _ in range(3):
        cur = F.max_pool2d(cur, kernel_size=2, stride=2)
        pooled.append(cur)
    out = []
    for p in pooled:
        cnt = torch.sum(torch.sum(p != 0, dtype=torch.int16, dim=[2]), dtype=torch.int16, dim=-1)
        out.append(x.shape[0] / p.numel() * cnt.float())
    return torch.stack(out, dim=1)


# --------------------------------------------------------------------------- a3
def position_embedding_sine(H: int, W: int, C: int, temperature: float = 10000.0) -> Tensor:
    """sast_rnn.py:180-213 with num_pos_feats=C/2, normalize=True, scale=2pi.  -> (1,H,W,C)."""
    nf = C // 2
    ones = torch.ones(1, H, W, dtype=torch.bool)
    ye = ones.cumsum(1, dtype=torch.float32)
    xe = ones.cumsum(2, dtype=torch.float32)
    eps, scale = 1e-6, 2 * math.pi
    ye = (ye - 0.5) / (ye[:, -1:, :] + eps) * scale
    xe = (xe - 0.5) / (xe[:, :, -1:] + eps) * scale
    dim_t = torch.arange(nf, dtype=torch.float32)
    dim_t = temperature ** (2 * (dim_t // 2) / nf)
    px = xe[:, :, :, None] / dim_t
    py = ye[:, :, :, None] / dim_t
    px = torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), dim=4).flatten(3)
    py = torch.stack((py[..., 0::2].sin(), py[..., 1::2].cos()), dim=4).flatten(3)
    return torch.cat((py, px), dim=3)


# --------------------------------------------------------------------------- a4
def window_partition(x: Tensor, hw: Tuple[int, int]) -> Tensor:
    """ops.py:189-195.  (B,H,W,C) -> (B*N,h,w,C), windows are contiguous h x w tiles."""
    B, H, W, C = x.shape
    h, w = hw
    assert H % h == 0 and W % w == 0
    return x.reshape(B, H // h, h, W // w, w, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, h, w, C)


def window_reverse(win: Tensor, hw: Tuple[int, int], img: Tuple[int, int]) -> Tensor:
    """ops.py:198-203."""
    H, W = img
    h, w = hw
    C = win.shape[-1]
    return win.reshape(-1, H // h, W // w, h, w, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, H, W, C)


def grid_partition(x: Tensor, hw: Tuple[int, int]) -> Tensor:
    """ops.py:206-212.  A grid group takes one token from each of the h x w coarse cells."""
    B, H, W, C = x.shape
    h, w = hw
    assert H % h == 0 and W % w == 0
    return x.reshape(B, h, H // h, w, W // w, C).permute(0, 2, 4, 1, 3, 5).reshape(-1, h, w, C)


def grid_reverse(win: Tensor, hw: Tuple[int, int], img: Tuple[int, int]) -> Tensor:
    """ops.py:215-220."""
    H, W = img
    h, w = hw
    C = win.shape[-1]
    return win.reshape(-1, H // h, W // w, h, w, C).permute(0, 3, 1, 4, 2, 5).reshape(-1, H, W, C)


# --------------------------------------------------------------------------- a2
def conv_downsample_cf2cl(x: Tensor, p: Params, pre: str, factor: int) -> Tensor:
    """ops.py:54-95: conv k=2f-1, stride f, replicate pad f-1, no bias -> NHWC -> LayerNorm(1e-5).  downsample_cfg.overlap False
    (:74-76) <=> the weight is f x f: no padding; norm_affine False (:69,87) <=> the dict holds no norm.weight / norm.bias."""
    w = p[pre + "conv.weight"]
    pad = w.shape[-1] // 2 if w.shape[-1] != factor else 0
    y = F.conv2d(F.pad(x, (pad, pad, pad, pad), mode="replicate") if pad else x, w, None, stride=factor)
    y = y.permute(0, 2, 3, 1).contiguous()
    return F.layer_norm(y, (y.shape[-1],), p.get(pre + "norm.weight"), p.get(pre + "norm.bias"), 1e-5)


# --------------------------------------------------------------------------- a6-a8
def select_windows(scores: Tensor, B: int, N: int, T: int, bounce: float) -> Tensor:
    """SAST.py:84-89 + :258-267.  scores (B,N,T,C) -> flat ascending window ids (M,)."""
    nw = (torch.norm(scores, dim=[2, 3], p=1) / T).softmax(-1).view(B, N)
    thr = (1 / N) / (1 + bounce)
    nz = torch.nonzero(nw >= thr)
    if B == 1:
        return nz[:, 1]
    return nz[:, 0] * N + nz[:, 1]


def select_tokens(scores: Tensor, index_window: Tensor, B: int, N: int, T: int, bounce: float):
    """SAST.py:91-96 + :270-281.  -> index_token (M*Kmax,), asy_index (sum K,), K (M,)."""
    nt = torch.norm(scores, dim=[3], p=1).view(B * N, -1)[index_window].softmax(-1)
    thr = (1 / T) / (1 + bounce)
    gt = nt >= thr
    K = gt.sum(dim=1)
    top = torch.topk(nt, k=int(K.max()), dim=1, largest=True, sorted=False)[1]
    base = torch.arange(0, nt.shape[0] * nt.shape[1], nt.shape[1]).view(-1, 1)
    nz = torch.nonzero(gt)
    asy = nz[:, 0] * nt.shape[1] + nz[:, 1]
    return (top + base).view(-1), asy, K


def selection_margins(scores: Tensor, B: int, N: int, T: int, bounce: float):
    """relative distance of every softmax value to its threshold; test diagnostics."""
    nw = (torch.norm(scores, dim=[2, 3], p=1) / T).softmax(-1).view(B, N)
    thr_n = torch.tensor((1 / N) / (1 + bounce), dtype=nw.dtype)
    nt = torch.norm(scores, dim=[3], p=1).view(B * N, -1).softmax(-1)
    thr_t = torch.tensor((1 / T) / (1 + bounce), dtype=nt.dtype)
    return ((nw - thr_n).abs() / thr_n), ((nt - thr_t).abs() / thr_t)


# --------------------------------------------------------------------------- a9
# layers/create_act.py:62-79 (_ACT_LAYER_DEFAULT; torch >= 1.7 has nn.SiLU): the names the build implements
def _hard_mish(x):
    """layers/activations.py:104-112 (the memory-efficient variant get_act_layer picks, activations_me.py, has the same derivative
    except on the two kinks)"""
    return 0.5 * x * (x + 2).clamp(min=0, max=2)


# every parameter-free name of get_act_layer (layers/create_act.py:62-98), mapped as the reference maps it on this torch: native
# F.silu / F.mish / F.hardsigmoid / F.hardswish, module defaults for leaky_relu (0.01), elu / celu (alpha 1), selu
GLU_ACTS = {"gelu": F.gelu, "relu": F.relu, "silu": F.silu, "swish": F.silu, "sigmoid": torch.sigmoid, "tanh": torch.tanh,
            "mish": F.mish, "relu6": F.relu6, "leaky_relu": F.leaky_relu, "elu": F.elu, "celu": F.celu, "selu": F.selu,
            "hard_sigmoid": F.hardsigmoid, "hardsigmoid": F.hardsigmoid, "hard_swish": F.hardswish, "hardswish": F.hardswish,
            "hard_mish": _hard_mish}


def drop_path(x: Tensor, cfg: "AttnCfg") -> Tensor:
    """layers/drop.py:137-154 as MS_WSA applies it (SAST.py:232,248) to (kept rows, C): one Bernoulli(keep_prob) factor per row, divided by
    keep_prob.  The same torch calls as the reference: under the same RNG state it draws the same factors."""
    if cfg.drop_path == 0.0 or not cfg.training:
        return x
    keep = 1.0 - cfg.drop_path
    if cfg.drop_masks is not None:
        rt = cfg.drop_masks.pop(0).view(-1, 1).to(x.dtype)
    else:
        rt = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        if keep > 0.0:
            rt.div_(keep)
    if cfg.drop_log is not None:
        cfg.drop_log.append(rt.view(-1).clone())
    return x * rt


def mlp_glu(x: Tensor, p: Params, pre: str, act: str = "gelu", cfg: Optional["AttnCfg"] = None) -> Tensor:
    """ops.py:111-175: Linear(C->2*inner) -> value * act(gate) (GELU_erf in every shipped config) -> Dropout(drop_mlp) -> Linear(inner->C).
    The dropout is the reference's torch call (same RNG draw); the mask goes through the same test plumbing as DropPath's factors."""
    y = F.linear(x, p[pre + "net.0.proj.weight"], p.get(pre + "net.0.proj.bias"))       # (mlp_bias: False -> no bias keys)
    val, gate = torch.tensor_split(y, 2, dim=-1)
    if act == "prelu":      # layers/activations.py:124-131: nn.PReLU with ONE learnable slope (init 0.25), `net.0.act_layer.weight`
        h = val * F.prelu(gate, p[pre + "net.0.act_layer.weight"])
    else:
        h = val * GLU_ACTS[act](gate)
    if cfg is not None and cfg.drop_mlp > 0.0 and cfg.training:
        if cfg.drop_masks is not None:
            mask = cfg.drop_masks.pop(0).view_as(h)
        else:
            mask = F.dropout(torch.ones_like(h), cfg.drop_mlp, True)      # keep / (1 - p): h * mask == F.dropout(h) bit for bit
        if cfg.drop_log is not None:
            cfg.drop_log.append(mask.clone())
        h = h * mask
    return F.linear(h, p[pre + "net.2.weight"], p.get(pre + "net.2.bias"))


def ms_wsa(x: Tensor, idx: Sequence[Tensor], B: int, p: Params, pre: str, cfg: AttnCfg) -> Tensor:
    """SAST.py:199-255 (padded top-k formulation, column mask -1e4).  x (B*N,T,C) -> same."""
    index_window, index_token, padding_index, asy_index, _K = idx
    M = len(index_window)
    shape = x.shape
    Nw, C = x.shape[0], x.shape[-1]
    heads, dh = C // cfg.dim_head, cfg.dim_head
    x = F.layer_norm(x.view(Nw, -1, C), (C,), p[pre + "norm1.weight"], p[pre + "norm1.bias"], cfg.norm_eps)
    if len(index_token) == 0:
        return x.view(*shape)
    X = x.clone()
    x = x[index_window].view(-1, C)
    XX = x.clone()
    x[asy_index] = F.layer_norm(x[asy_index], (C,), p[pre + "norm2.weight"], p[pre + "norm2.bias"], cfg.norm_eps)
    shortcut = x[asy_index]
    x = x[index_token].view(M, -1, C)

    qkv = F.linear(x, p[pre + "qkv.weight"], p.get(pre + "qkv.bias"))                  # (attention_bias: False -> no bias keys)
    q, k, v = qkv.view(M, -1, heads, dh * 3).transpose(1, 2).chunk(3, dim=3)
    attn = (q @ k.transpose(-2, -1)) * (dh ** -0.5)
    Kmax = q.shape[2]
    amap = torch.zeros((XX.shape[0], Kmax, heads), dtype=attn.dtype)
    amap[index_token] = attn.transpose(1, 3).reshape(-1, Kmax, heads)
    amap[padding_index] = -1e4
    attn = amap[index_token].view(M, -1, Kmax, heads).transpose(1, 3)
    attn = attn.softmax(dim=-1)
    x = (attn @ v).transpose(1, 2)
    x = F.linear(x.reshape(M, -1, C), p[pre + "proj.weight"], p.get(pre + "proj.bias"))

    XX[index_token] = x.view(-1, C)
    x = XX[asy_index]
    x = shortcut + drop_path(x * p[pre + "ls1.gamma"], cfg)
    shortcut = x
    x = mlp_glu(x, p, pre + "mlp.", cfg.mlp_activation, cfg)
    if cfg.enable_cb:  # SAST.py:240-246
        tX, tXX = torch.zeros_like(X), torch.zeros_like(XX)
        tXX[asy_index] = x
        tX[index_window] = tXX.view(M, -1, C)
        tX = tX.view(B, -1, C)
        tX = (0.5 * tX + (1 - 0.5) * tX.mean(dim=1, keepdim=True)).view(*shape)
        x = tX[index_window].view(-1, C)[asy_index]
    x = shortcut + drop_path(x * p[pre + "ls2.gamma"], cfg)
    XX[asy_index] = x.view(-1, C)
    XX[padding_index] = X[index_window].view(-1, C)[padding_index]
    X[index_window] = XX.view(M, -1, C)
    return X.view(*shape)


# --------------------------------------------------------------------------- a5,a10
def _padding_index(index_token: Tensor, asy_index: Tensor) -> Tensor:
    return index_token[torch.isin(index_token, asy_index, assume_unique=True, invert=True)]


def selection_diff(scores: Tensor, own, other, B: int, N: int, T: int, bounce: float) -> dict:
    """test diagnostics for full-size runs (token decisions sit within 1 ulp of the threshold, so two
    correct fp32 implementations with different summation orders may disagree on a handful of them): the window /
    token decisions on which two index lists differ, with the relative threshold margin of each (from `scores`, this
    side's own values).  Tokens of a window only one side kept are attributed to the window decision."""
    mw, mt = selection_margins(scores, B, N, T, bounce)
    mw, mt = mw.reshape(-1), mt.reshape(B * N, T)
    iw_a, iw_b = own[0], other[0]
    wa, wb = torch.zeros(B * N, dtype=torch.bool), torch.zeros(B * N, dtype=torch.bool)
    wa[iw_a] = True
    wb[iw_b] = True
    wdiff = wa ^ wb
    ta, tb = torch.zeros(B * N, T, dtype=torch.bool), torch.zeros(B * N, T, dtype=torch.bool)
    ta.view(-1)[iw_a[torch.div(own[3], T, rounding_mode="floor")] * T + own[3] % T] = True
    tb.view(-1)[iw_b[torch.div(other[3], T, rounding_mode="floor")] * T + other[3] % T] = True
    tdiff = (ta ^ tb) & ~wdiff[:, None]
    return {"win_diff": int(wdiff.sum()), "tok_diff": int(tdiff.sum()), "decisions": int(B * N + wa.sum() * T),
            "max_margin": float(torch.cat([mw[wdiff], mt[tdiff], torch.zeros(1, dtype=mw.dtype)]).max())}


def _margin_entry(scores: Tensor, index_window: Tensor, B: int, N: int, T: int, bounce: float, where: str) -> dict:
    mw, mt = selection_margins(scores, B, N, T, bounce)
    flat = index_window if B > 1 else index_window      # (B == 1: ids are already 0..N-1)
    mt = mt[flat]
    allm = torch.cat([mw.reshape(-1), mt.reshape(-1)])
    return {"where": where, "win_min": float(mw.min()), "tok_min": float(mt.min()),
            "below": {k: int((allm < float(k)).sum()) for k in ("1e-7", "1e-6", "1e-5", "1e-4")}, "decisions": int(allm.numel())}


def sast_block(x: Tensor, pe: Tensor, r: Tensor, p: Params, pre: str, cfg: AttnCfg,
               index_list=None, first_block: bool = True, return_scores: bool = False, forced_lists=None, diff_log=None,
               kink_log: Optional[dict] = None, margin_log: Optional[list] = None):
    """SAST.py:98-164.  x (B,H,W,C) NHWC, pe (1,H,W,C), r (B,20) -> (x, index_count, [list1,list2]).
    forced_lists (test diagnostics, never the reference's behaviour): [list1, list2] to USE instead of this block's own
    selection; the own selection is still computed and its disagreement with the forced one is appended to diff_log.
    kink_log (test diagnostics): kink_log[pre] = {"x": input of the scoring linear, "z": its pre-activation, "s": ReLU(z) with
    retain_grad} -- what a kink-aware comparison of the `to_scores` gradients needs (tests/parity_helpers.py:scores_grads_close).
    margin_log (fixture generation / diagnostics): per selection the relative distance of the DECIDED softmax values (all windows; the
    tokens of the kept windows) to their thresholds: {"where", "win_min", "tok_min", "below": counts under 1e-7 / 1e-6 / 1e-5 / 1e-4}."""
    B, H, W, C = x.shape
    h, w = cfg.partition_size
    T = h * w
    N = H * W // T
    count = 0
    x = x + pe[:, :H, :W, :].repeat(B, 1, 1, 1)
    x = window_partition(x, (h, w)).view(B, N, -1, C)
    scores = None
    if first_block:
        scale = F.linear(r + 1e-6, torch.exp(p[pre + "to_controls.weight"]))[:, None, None, :]
        z_pre = F.linear(x, p[pre + "to_scores.weight"], p[pre + "to_scores.bias"])
        scores = F.relu(z_pre)
        if kink_log is not None:
            if scores.requires_grad:
                scores.retain_grad()
            kink_log.setdefault(pre, []).append({"x": x, "z": z_pre, "s": scores})
        weight = scale.sigmoid() * scores.sigmoid()
        x = (weight * x).view(B * N, -1, C)
        scale = cfg.amp / scale
        scale[scale == torch.inf] = 0
        scores = scale * scores
        iw = select_windows(scores, B, N, T, cfg.bounce)
        it, asy, K = select_tokens(scores, iw, B, N, T, cfg.bounce)
        list1 = [iw, it, _padding_index(it, asy), asy, K]
        if margin_log is not None:
            margin_log.append(_margin_entry(scores.detach(), iw, B, N, T, cfg.bounce, pre + "win"))
        if forced_lists is not None:
            if diff_log is not None:
                diff_log.append(dict(selection_diff(scores.detach(), list1, forced_lists[0], B, N, T, cfg.bounce), where=pre + "win"))
            list1 = forced_lists[0]
    else:
        x = x.view(B * N, -1, C)
        list1, list2 = index_list
    if len(list1[1]):
        x = ms_wsa(x, list1, B, p, pre + "win_attn.", cfg)
    x = window_reverse(x, (h, w), (H, W))
    count += len(list1[3]) // B
    scores_win = scores
    if first_block:
        scores = window_reverse(scores.view_as(x), (h, w), (H, W))
        scores = grid_partition(scores, (h, w)).view(B, N, -1, C)
        iw = select_windows(scores, B, N, T, cfg.bounce)
        it, asy, K = select_tokens(scores, iw, B, N, T, cfg.bounce)
        list2 = [iw, it, _padding_index(it, asy), asy, K]
        if margin_log is not None:
            margin_log.append(_margin_entry(scores.detach(), iw, B, N, T, cfg.bounce, pre + "grid"))
        if forced_lists is not None:
            if diff_log is not None:
                diff_log.append(dict(selection_diff(scores.detach(), list2, forced_lists[1], B, N, T, cfg.bounce), where=pre + "grid"))
            list2 = forced_lists[1]
    x = grid_partition(x.view(B, H, W, C), (h, w)).view(B * N, -1, C)
    if len(list2[1]):
        x = ms_wsa(x, list2, B, p, pre + "grid_attn.", cfg)
    x = grid_reverse(x, (h, w), (H, W))
    count += len(list2[3]) // B
    if return_scores:
        return x, count, [list1, list2], scores_win
    return x, count, [list1, list2]


# --------------------------------------------------------------------------- a12
def conv_lstm(x: Tensor, hc: Optional[Tuple[Tensor, Tensor]], p: Params, pre: str, cell_update_dropout: float = 0.0, training: bool = True,
              drop_mask: Optional[Tensor] = None):
    """rnn.py:36-69.  NCHW.  dws_conv=True <=> the dict holds `<pre>conv3x3_dws.weight`: [C,1,k,k] = on the previous hidden state
    (dws_conv_only_hidden=True, :52-53), [2C,1,k,k] = on cat(x, h) (:55-56).  cell_update_dropout (:34,64): nn.Dropout on the tanh of
    the cell input -- the same torch call, so under the same RNG state it draws the reference's mask; drop_mask (NCHW, keep / (1 - p))
    replaces the draw (fixture tests/golden/lstm_dropout.npz)."""
    C = x.shape[1]
    if hc is None:
        hc = (torch.zeros_like(x), torch.zeros_like(x))
    h0, c0 = hc
    wd = p.get(pre + "conv3x3_dws.weight")
    if wd is not None and wd.shape[0] == C:
        h0 = F.conv2d(h0, wd, p.get(pre + "conv3x3_dws.bias"), padding=wd.shape[-1] // 2, groups=C)
    xh = torch.cat((x, h0), dim=1)
    if wd is not None and wd.shape[0] == 2 * C:
        xh = F.conv2d(xh, wd, p.get(pre + "conv3x3_dws.bias"), padding=wd.shape[-1] // 2, groups=2 * C)
    mix = F.conv2d(xh, p[pre + "conv1x1.weight"], p[pre + "conv1x1.bias"])
    gates, cin = torch.tensor_split(mix, [C * 3], dim=1)
    f, i, o = torch.tensor_split(torch.sigmoid(gates), 3, dim=1)
    cell_in = torch.tanh(cin) * drop_mask if drop_mask is not None else F.dropout(torch.tanh(cin), cell_update_dropout, training)
    c1 = f * c0 + i * cell_in
    h1 = o * torch.tanh(c1)
    return h1, c1


# --------------------------------------------------------------------------- a11
def backbone_stage(x: Tensor, state, r: Tensor, p: Params, pre: str, cfg: BackboneCfg, stage_idx: int,
                   pe: Optional[Tensor] = None, token_mask: Optional[Tensor] = None, forced_lists=None, diff_log=None, kink_log=None,
                   margin_log=None):
    """sast_rnn.py:265-287.  NCHW in -> (h NCHW, (h,c), P, index lists).  token_mask (B,H,W) bool: x[token_mask] = mask_token
    (:271-273, parameter `<pre>mask_token` of shape (1,1,1,C), only stage 0 has one when enable_masking is set)."""
    factor = cfg.patch_size if stage_idx == 0 else 2
    x = conv_downsample_cf2cl(x, p, pre + "downsample_cf2cl.", factor)
    if token_mask is not None:
        x = x.clone()
        x[token_mask] = p[pre + "mask_token"].to(x.dtype)
    B, H, W, C = x.shape
    if pe is None:
        pe = position_embedding_sine(H, W, C).to(x.dtype)
    P = 0
    lists = None
    all_lists = []
    for bi in range(cfg.num_blocks[stage_idx]):
        x, cnt, lists = sast_block(x, pe, r, p, f"{pre}att_blocks.{bi}.att.", cfg.attn,
                                   index_list=lists, first_block=(bi == 0),
                                   forced_lists=forced_lists[bi] if (forced_lists is not None and bi == 0) else None, diff_log=diff_log,
                                   kink_log=kink_log, margin_log=margin_log)
        all_lists.append(lists)
        P += cnt
    x = x.permute(0, 3, 1, 2).contiguous()
    hc = conv_lstm(x, state, p, pre + "lstm.")
    return hc[0], hc, P, all_lists


def backbone(x: Tensor, prev_states, p: Params, cfg: BackboneCfg, pre: str = "", return_lists: bool = False,
             token_mask: Optional[Tensor] = None, forced_lists=None, diff_log=None, kink_log=None, margin_log=None):
    """sast_rnn.py:144-162.  x (B,20,H,W) -> ({1..4: h}, states, P).
    forced_lists[stage][block] = [list1, list2] / diff_log: see sast_block (full-size parity diagnostics only)."""
    if prev_states is None:
        prev_states = [None] * 4
    r = non_zero_ratio(x)
    if x.dtype == torch.float64:       # fp64 arbiter run: same graph in double, parameters given in double
        r = r.double()
    else:
        x = x.float()
    out, states, P, lists = {}, [], [], []
    for s in range(4):
        x, st, cnt, ls = backbone_stage(x, prev_states[s], r[:, s], p, f"{pre}stages.{s}.", cfg, s,
                                        token_mask=token_mask if s == 0 else None,
                                        forced_lists=forced_lists[s] if forced_lists is not None else None, diff_log=diff_log,
                                        kink_log=kink_log, margin_log=margin_log)
        states.append(st)
        out[s + 1] = st[0]
        P.append(cnt)
        lists.append(ls)
    if return_lists:
        return out, states, P, lists
    return out, states, P


# --------------------------------------------------------------------------- a13
def base_conv(x: Tensor, p: Params, pre: str, stride: int, training: bool, bufs: Optional[Params] = None):
    """network_blocks.py:29-54: conv(no bias, same pad) + BatchNorm2d(eps 1e-5, mom 0.1) + SiLU."""
    w = p[pre + "conv.weight"]
    y = F.conv2d(x, w, None, stride=stride, padding=(w.shape[-1] - 1) // 2, groups=x.shape[1] // w.shape[1])   # groups > 1: DWConv.dconv
    src = bufs if bufs is not None else p
    rm, rv = src.get(pre + "bn.running_mean"), src.get(pre + "bn.running_var")
    if training and bufs is None:  # do not mutate the caller's buffers unless asked to
        rm = rm.clone() if rm is not None else None
        rv = rv.clone() if rv is not None else None
    y = F.batch_norm(y, rm, rv, p[pre + "bn.weight"], p[pre + "bn.bias"], training, 0.1, 1e-5)
    return F.silu(y)


def conv_unit(x: Tensor, p: Params, pre: str, stride: int, training: bool, bufs: Optional[Params] = None):
    """`Conv = DWConv if depthwise else BaseConv` (network_blocks.py:93, yolo_pafpn.py:37, yolo_head.py:42): which one a unit is follows
    from its parameter names -- DWConv (network_blocks.py:57-76) holds `dconv` (depth-wise, carries the stride) and `pconv` (1x1)."""
    if pre + "dconv.conv.weight" in p:
        return base_conv(base_conv(x, p, pre + "dconv.", stride, training, bufs), p, pre + "pconv.", 1, training, bufs)
    return base_conv(x, p, pre, stride, training, bufs)


def csp_layer(x: Tensor, p: Params, pre: str, n: int, training: bool, bufs=None):
    """network_blocks.py:104-141 with shortcut=False, expansion 0.5, Bottleneck expansion 1.0."""
    x1 = base_conv(x, p, pre + "conv1.", 1, training, bufs)
    x2 = base_conv(x, p, pre + "conv2.", 1, training, bufs)
    for i in range(n):
        x1 = conv_unit(base_conv(x1, p, f"{pre}m.{i}.conv1.", 1, training, bufs), p,
                       f"{pre}m.{i}.conv2.", 1, training, bufs)
    return base_conv(torch.cat((x1, x2), dim=1), p, pre + "conv3.", 1, training, bufs)


def pafpn(feats: Dict[int, Tensor], p: Params, pre: str = "", depth: float = 0.67,
          in_stages=(2, 3, 4), training: bool = True, bufs=None):
    """yolo_pafpn.py:109-139."""
    n = round(3 * depth)
    x2, x1, x0 = (feats[s] for s in in_stages)
    up = lambda t: F.interpolate(t, scale_factor=2, mode="nearest-exact")
    fpn0 = base_conv(x0, p, pre + "lateral_conv0.", 1, training, bufs)
    f0 = csp_layer(torch.cat([up(fpn0), x1], 1), p, pre + "C3_p4.", n, training, bufs)
    fpn1 = base_conv(f0, p, pre + "reduce_conv1.", 1, training, bufs)
    pan2 = csp_layer(torch.cat([up(fpn1), x2], 1), p, pre + "C3_p3.", n, training, bufs)
    d1 = conv_unit(pan2, p, pre + "bu_conv2.", 2, training, bufs)
    pan1 = csp_layer(torch.cat([d1, fpn1], 1), p, pre + "C3_n3.", n, training, bufs)
    d0 = conv_unit(pan1, p, pre + "bu_conv1.", 2, training, bufs)
    pan0 = csp_layer(torch.cat([d0, fpn0], 1), p, pre + "C3_n4.", n, training, bufs)
    return pan2, pan1, pan0


# --------------------------------------------------------------------------- YOLOX head, inference path (forward only)
def head_hidden_dim(in_channels) -> int:
    """yolo_head.py:48-56: width = in_channels[-1] / 1024, hidden = int(256 * width)."""
    return int(256 * (in_channels[-1] / 1024))


def yolox_head_eval(feats: Sequence[Tensor], p: Params, strides=(8, 16, 32), pre: str = "", decode: bool = True) -> Tensor:
    """YOLOXHead.forward in eval mode (yolo_head.py:165-246 without the training branch) + decode_outputs (:264-289).
    feats: the three PAFPN maps (B,C,H,W), strides ascending -> (B, sum H*W, 5 + num_classes):
    [cx, cy, w, h, sigmoid(obj), sigmoid(cls...)] (decoded: (xy + grid) * stride, exp(wh) * stride)."""
    outs, grids, strs = [], [], []
    for k, (x, stride) in enumerate(zip(feats, strides)):
        x = base_conv(x, p, f"{pre}stems.{k}.", 1, False)
        cf = x
        rf = x
        for i in range(2):
            cf = conv_unit(cf, p, f"{pre}cls_convs.{k}.{i}.", 1, False)
            rf = conv_unit(rf, p, f"{pre}reg_convs.{k}.{i}.", 1, False)
        cls = F.conv2d(cf, p[f"{pre}cls_preds.{k}.weight"], p[f"{pre}cls_preds.{k}.bias"])
        reg = F.conv2d(rf, p[f"{pre}reg_preds.{k}.weight"], p[f"{pre}reg_preds.{k}.bias"])
        obj = F.conv2d(rf, p[f"{pre}obj_preds.{k}.weight"], p[f"{pre}obj_preds.{k}.bias"])
        o = torch.cat([reg, obj.sigmoid(), cls.sigmoid()], 1)
        H, W = o.shape[-2:]
        outs.append(o.flatten(start_dim=2))
        yv, xv = torch.meshgrid([torch.arange(H, dtype=o.dtype), torch.arange(W, dtype=o.dtype)], indexing="ij")
        grids.append(torch.stack((xv, yv), 2).view(1, -1, 2))
        strs.append(torch.full((1, H * W, 1), float(stride), dtype=o.dtype))
    out = torch.cat(outs, dim=2).permute(0, 2, 1)
    if not decode:
        return out
    g, st = torch.cat(grids, 1), torch.cat(strs, 1)
    return torch.cat([(out[..., 0:2] + g) * st, torch.exp(out[..., 2:4]) * st, out[..., 4:]], dim=-1)


# ---- YOLOX head, training branch (yolo_head.py:165-246 training path, get_losses :291-443, SimOTA :452-606, IOUloss losses.py:10-55)
def _bboxes_iou_cxcywh(a: Tensor, b: Tensor) -> Tensor:
    """yolox/utils/boxes.py:79-103 with xyxy=False: pairwise IoU of (cx,cy,w,h) boxes, a (G,4) x b (N,4) -> (G,N)."""
    tl = torch.max(a[:, None, :2] - a[:, None, 2:] / 2, b[:, :2] - b[:, 2:] / 2)
    br = torch.min(a[:, None, :2] + a[:, None, 2:] / 2, b[:, :2] + b[:, 2:] / 2)
    area_a, area_b = a[:, 2] * a[:, 3], b[:, 2] * b[:, 3]
    en = (tl < br).type(tl.type())
    en = en[:, :, 0] * en[:, :, 1]
    area_i = torch.prod(br - tl, 2) * en
    return area_i / (area_a[:, None] + area_b - area_i)


def _iou_loss(pred: Tensor, target: Tensor) -> Tensor:
    """losses.py:16-33 (loss_type "iou", reduction none): 1 - iou^2."""
    tl = torch.max(pred[:, :2] - pred[:, 2:] / 2, target[:, :2] - target[:, 2:] / 2)
    br = torch.min(pred[:, :2] + pred[:, 2:] / 2, target[:, :2] + target[:, 2:] / 2)
    area_p, area_g = torch.prod(pred[:, 2:], 1), torch.prod(target[:, 2:], 1)
    en = (tl < br).type(tl.type()).prod(dim=1)
    area_i = torch.prod(br - tl, 1) * en
    iou = area_i / (area_p + area_g - area_i + 1e-16)
    return 1 - iou ** 2


def simota_assign(gt_boxes: Tensor, gt_classes: Tensor, boxes: Tensor, cls_logits: Tensor, obj_logits: Tensor, strides: Tensor,
                  xs: Tensor, ys: Tensor, num_classes: int):
    """get_assignments (yolo_head.py:452-538) for ONE image: geometry constraint (:540-571, centre radius 1.5 strides), cost =
    BCE(sqrt(sigmoid(cls) sigmoid(obj)), one_hot) + 3 * (-log(iou + 1e-8)) + 1e6 * not-in-centre, dynamic-k matching (:573-606).
    boxes (A,4) decoded predictions, cls_logits (A,nc), obj_logits (A,1), strides / xs / ys (A,).
    -> fg_mask (A,) bool, matched_gt_inds (num_fg,), pred_ious_this_matching (num_fg,), gt_matched_classes (num_fg,)"""
    G = gt_boxes.shape[0]
    xc, yc = ((xs + 0.5) * strides).unsqueeze(0), ((ys + 0.5) * strides).unsqueeze(0)
    dist = strides.unsqueeze(0) * 1.5
    c_l = xc - (gt_boxes[:, 0:1] - dist)
    c_r = (gt_boxes[:, 0:1] + dist) - xc
    c_t = yc - (gt_boxes[:, 1:2] - dist)
    c_b = (gt_boxes[:, 1:2] + dist) - yc
    is_in = torch.stack([c_l, c_t, c_r, c_b], 2).min(dim=-1).values > 0.0
    fg_mask = is_in.sum(dim=0) > 0
    geometry = is_in[:, fg_mask]
    b_fg, cls_fg, obj_fg = boxes[fg_mask], cls_logits[fg_mask], obj_logits[fg_mask]
    n_in = b_fg.shape[0]
    ious = _bboxes_iou_cxcywh(gt_boxes, b_fg)
    onehot = F.one_hot(gt_classes.to(torch.int64), num_classes).float()
    iou_cost = -torch.log(ious + 1e-8)
    p = (cls_fg.float().sigmoid() * obj_fg.float().sigmoid()).sqrt()
    cls_cost = F.binary_cross_entropy(p.unsqueeze(0).repeat(G, 1, 1), onehot.unsqueeze(1).repeat(1, n_in, 1), reduction="none").sum(-1)
    cost = cls_cost + 3.0 * iou_cost + float(1e6) * (~geometry)
    matching = torch.zeros_like(cost, dtype=torch.uint8)
    k_cand = min(10, ious.size(1))
    topk_ious, _ = torch.topk(ious, k_cand, dim=1)
    dynamic_ks = torch.clamp(topk_ious.sum(1).int(), min=1)
    for g in range(G):
        _, pos = torch.topk(cost[g], k=int(dynamic_ks[g]), largest=False)
        matching[g][pos] = 1
    per_anchor = matching.sum(0)
    if per_anchor.max() > 1:
        multi = per_anchor > 1
        _, amin = torch.min(cost[:, multi], dim=0)
        matching[:, multi] *= 0
        matching[amin, multi] = 1
    fg_in = per_anchor > 0
    fg_mask = fg_mask.clone()
    fg_mask[fg_mask.clone()] = fg_in
    matched = matching[:, fg_in].argmax(0)
    pred_ious = (matching * ious).sum(0)[fg_in]
    return fg_mask, matched, pred_ious, gt_classes[matched]


def yolox_head_train(feats: Sequence[Tensor], labels: Tensor, p: Params, strides=(8, 16, 32), pre: str = "", num_classes: int = 3,
                     bufs: Optional[Params] = None, use_l1: bool = False):
    """YOLOXHead.forward in TRAINING mode (batch-statistics BatchNorm) + get_losses: -> dict(loss, iou_loss, conf_loss, cls_loss,
    num_fg) and the assignment of every image (for index-level parity checks).  labels (B, max_labels, 5) = (cls, cx, cy, w, h),
    valid rows first, rows that sum to 0 are padding (yolo_head.py:306)."""
    outs, xs, ys, ss, origin = [], [], [], [], []
    for k, (x, stride) in enumerate(zip(feats, strides)):
        x = base_conv(x, p, f"{pre}stems.{k}.", 1, True, bufs)
        cf, rf = x, x
        for i in range(2):
            cf = conv_unit(cf, p, f"{pre}cls_convs.{k}.{i}.", 1, True, bufs)
            rf = conv_unit(rf, p, f"{pre}reg_convs.{k}.{i}.", 1, True, bufs)
        cls = F.conv2d(cf, p[f"{pre}cls_preds.{k}.weight"], p[f"{pre}cls_preds.{k}.bias"])
        reg = F.conv2d(rf, p[f"{pre}reg_preds.{k}.weight"], p[f"{pre}reg_preds.{k}.bias"])
        obj = F.conv2d(rf, p[f"{pre}obj_preds.{k}.weight"], p[f"{pre}obj_preds.{k}.bias"])
        o = torch.cat([reg, obj, cls], 1)
        B, no, H, W = o.shape
        origin.append(reg.view(B, 4, H * W).permute(0, 2, 1))      # raw regression outputs (use_l1, :199-208)
        o = o.view(B, no, H * W).permute(0, 2, 1)
        yv, xv = torch.meshgrid([torch.arange(H, dtype=o.dtype), torch.arange(W, dtype=o.dtype)], indexing="ij")
        grid = torch.stack((xv, yv), 2).view(1, -1, 2)
        o = torch.cat([(o[..., :2] + grid) * stride, torch.exp(o[..., 2:4]) * stride, o[..., 4:]], dim=-1)   # get_output_and_grid :248-262
        outs.append(o)
        xs.append(grid[0, :, 0]); ys.append(grid[0, :, 1]); ss.append(torch.full((H * W,), float(stride), dtype=o.dtype))
    out = torch.cat(outs, 1)
    xs, ys, ss = torch.cat(xs), torch.cat(ys), torch.cat(ss)
    boxes, objp, clsp = out[:, :, :4], out[:, :, 4:5], out[:, :, 5:]
    nlabel = (labels.sum(dim=2) > 0).sum(dim=1)
    A = out.shape[1]
    origin = torch.cat(origin, 1)
    l1_t = []
    cls_t, reg_t, obj_t, fg_all, assigns = [], [], [], [], []
    num_fg, num_gts = 0.0, 0.0
    for b in range(out.shape[0]):
        G = int(nlabel[b])
        num_gts += G
        if G == 0:
            cls_t.append(out.new_zeros((0, num_classes))); reg_t.append(out.new_zeros((0, 4)))
            obj_t.append(out.new_zeros((A, 1))); fg_all.append(out.new_zeros(A).bool())
            assigns.append((fg_all[-1], torch.zeros(0, dtype=torch.long), out.new_zeros(0)))
            continue
        gtb, gtc = labels[b, :G, 1:5], labels[b, :G, 0]
        with torch.no_grad():
            fg, matched, pious, mcls = simota_assign(gtb, gtc, boxes[b], clsp[b], objp[b], ss, xs, ys, num_classes)
        num_fg += int(fg.sum())
        cls_t.append(F.one_hot(mcls.to(torch.int64), num_classes) * pious.unsqueeze(-1))
        obj_t.append(fg.unsqueeze(-1).to(out.dtype)); reg_t.append(gtb[matched]); fg_all.append(fg)
        gm, sm = gtb[matched], ss[fg]                                 # get_l1_target :445-450
        l1_t.append(torch.stack([gm[:, 0] / sm - xs[fg], gm[:, 1] / sm - ys[fg], torch.log(gm[:, 2] / sm + 1e-8),
                                 torch.log(gm[:, 3] / sm + 1e-8)], 1))
        assigns.append((fg, matched, pious))
    cls_t, reg_t, obj_t, fg_all = torch.cat(cls_t, 0), torch.cat(reg_t, 0), torch.cat(obj_t, 0), torch.cat(fg_all, 0)
    num_fg = max(num_fg, 1)
    loss_iou = _iou_loss(boxes.reshape(-1, 4)[fg_all], reg_t).sum() / num_fg
    loss_obj = F.binary_cross_entropy_with_logits(objp.reshape(-1, 1), obj_t, reduction="none").sum() / num_fg
    loss_cls = F.binary_cross_entropy_with_logits(clsp.reshape(-1, num_classes)[fg_all], cls_t, reduction="none").sum() / num_fg
    loss_l1 = (origin.reshape(-1, 4)[fg_all] - torch.cat(l1_t, 0)).abs().sum() / num_fg if (use_l1 and l1_t) else out.new_zeros(())
    loss = 5.0 * loss_iou + loss_obj + loss_cls + loss_l1
    return {"loss": loss, "iou_loss": 5.0 * loss_iou, "conf_loss": loss_obj, "cls_loss": loss_cls, "l1_loss": loss_l1,
            "num_fg": num_fg / max(num_gts, 1),
            "assign": assigns, "outputs": out}


# ---- post-processing: yolox/utils/boxes.py:32-76.  The reference calls torchvision.ops.nms / batched_nms
# (boxes.py:57-69); torchvision is not installed here and is not part of /root/reference, so its PUBLISHED algorithm (torchvision 0.15,
# the version the reference's README pins with pytorch 2.0: torchvision/ops/boxes.py `batched_nms`, csrc/ops/cpu/nms_kernel.cpp) is
# restated: parity for this function is pinned to that restatement, not to a run of torchvision ("parity unpinned").
#   nms: boxes by decreasing score; a box is kept unless an already kept box has IoU > thr with it, IoU = inter / (a_i + a_j - inter),
#        inter = max(0, min(x2) - max(x1)) * max(0, min(y2) - max(y1)), areas (x2 - x1) * (y2 - y1), all in fp32.
#   batched_nms: with at most 4000 box COORDINATES on the CPU (20000 on a GPU) the "coordinate trick": every box is shifted by
#        class * (max coordinate of all boxes + 1) and ONE class-agnostic nms runs on the shifted boxes -- the fp32 rounding of the
#        shifted corners moves intersections and areas by ~1e-7 relative to the unshifted per-class evaluation, which can flip a decision
#        that sits on the threshold; above that size a per-class loop over the unshifted boxes ("vanilla"), result sorted by score.
def _nms_greedy(boxes: Tensor, scores: Tensor, classes: Tensor, thr: float, class_agnostic: bool = False) -> Tensor:
    """plain greedy NMS on the boxes as given; class-aware = only boxes of the same class suppress each other (the vanilla form)"""
    order = torch.sort(scores, descending=True, stable=True).indices
    keep: List[int] = []
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    for i in order.tolist():
        ok = True
        for j in keep:
            if not class_agnostic and classes[i] != classes[j]:
                continue
            w = (torch.min(boxes[i, 2], boxes[j, 2]) - torch.max(boxes[i, 0], boxes[j, 0])).clamp(min=0)
            h = (torch.min(boxes[i, 3], boxes[j, 3]) - torch.max(boxes[i, 1], boxes[j, 1])).clamp(min=0)
            inter = w * h
            if inter / (area[i] + area[j] - inter) > thr:
                ok = False
                break
        if ok:
            keep.append(i)
    return torch.tensor(keep, dtype=torch.long)


BATCHED_NMS_TRICK_MAX_COORDS = 4000      # torchvision 0.15 batched_nms, CPU tensors (the oracle's device)


def _batched_nms(boxes: Tensor, scores: Tensor, classes: Tensor, thr: float) -> Tensor:
    """torchvision.ops.batched_nms as published (see above): coordinate trick up to 4000 coordinates, per-class loop beyond"""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.long)
    if boxes.numel() > BATCHED_NMS_TRICK_MAX_COORDS:
        return _nms_greedy(boxes, scores, classes, thr, class_agnostic=False)
    max_coordinate = boxes.max()
    offsets = classes.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
    return _nms_greedy(boxes + offsets[:, None], scores, classes, thr, class_agnostic=True)


def postprocess(prediction: Tensor, num_classes: int, conf_thre: float = 0.7, nms_thre: float = 0.45, class_agnostic: bool = False):
    """boxes.py:32-76 (class-aware branch, or torchvision.ops.nms over all boxes with class_agnostic): prediction (B, A, 5+nc) with (cx, cy, w, h, obj, cls...) -> list of (n_i, 7) tensors
    (x1, y1, x2, y2, obj_conf, class_conf, class_pred) sorted by decreasing score, or None."""
    pred = prediction.clone()
    pred[:, :, 0] = prediction[:, :, 0] - prediction[:, :, 2] / 2
    pred[:, :, 1] = prediction[:, :, 1] - prediction[:, :, 3] / 2
    pred[:, :, 2] = prediction[:, :, 0] + prediction[:, :, 2] / 2
    pred[:, :, 3] = prediction[:, :, 1] + prediction[:, :, 3] / 2
    out = [None] * pred.shape[0]
    for i, ip in enumerate(pred):
        class_conf, class_pred = torch.max(ip[:, 5:5 + num_classes], 1, keepdim=True)
        mask = (ip[:, 4] * class_conf.squeeze(1)) >= conf_thre
        det = torch.cat((ip[:, :5], class_conf, class_pred.float()), 1)[mask]
        if not det.size(0):
            continue
        if class_agnostic:
            keep = _nms_greedy(det[:, :4], det[:, 4] * det[:, 5], det[:, 6], nms_thre, True)      # torchvision.ops.nms (boxes.py:58-62)
        else:
            keep = _batched_nms(det[:, :4], det[:, 4] * det[:, 5], det[:, 6], nms_thre)            # torchvision.ops.batched_nms (:64-69)
        out[i] = det[keep]
    return out


def synthetic_labels(B: int, hw: Tuple[int, int], num_classes: int, max_labels: int = 8, seed: int = 0) -> Tensor:
    """(B, max_labels, 5) = (cls, cx, cy, w, h) in input pixels; a random number of valid rows first, zero rows after."""
    g = torch.Generator().manual_seed(seed)
    lab = torch.zeros(B, max_labels, 5)
    for b in range(B):
        n = int(torch.randint(0 if b == B - 1 and B > 2 else 1, max_labels + 1, (1,), generator=g))
        for i in range(n):
            w = float(torch.rand(1, generator=g)) * hw[1] * 0.3 + 8
            h = float(torch.rand(1, generator=g)) * hw[0] * 0.3 + 8
            cx = float(torch.rand(1, generator=g)) * (hw[1] - w) + w / 2
            cy = float(torch.rand(1, generator=g)) * (hw[0] - h) + h / 2
            lab[b, i] = torch.tensor([float(torch.randint(0, num_classes, (1,), generator=g)), cx, cy, w, h])
    return lab


def init_head_params(in_channels=(128, 256, 512), num_classes: int = 3, seed: int = 2, prior_prob: float = 0.01,
                     depthwise: bool = False) -> Params:
    """seeded stand-in for YOLOXHead.__init__ (names / shapes of yolo_head.py:58-133; biases as initialize_biases :154-163);
    BatchNorm running statistics are randomised so that the eval path is exercised with non-trivial statistics."""
    g = torch.Generator().manual_seed(seed)
    hid = head_hidden_dim(in_channels)
    p: Params = {}

    def conv_bn(name, ci, co, k):
        b = 1.0 / math.sqrt(ci * k * k)
        p[name + ".conv.weight"] = (torch.rand((co, ci, k, k), generator=g) * 2 - 1) * b
        p[name + ".bn.weight"] = 0.5 + torch.rand(co, generator=g)
        p[name + ".bn.bias"] = (torch.rand(co, generator=g) - 0.5) * 0.2
        p[name + ".bn.running_mean"] = (torch.rand(co, generator=g) - 0.5) * 0.2
        p[name + ".bn.running_var"] = 0.5 + torch.rand(co, generator=g)

    for k, ci in enumerate(in_channels):
        conv_bn(f"stems.{k}", ci, hid, 1)
        for i in range(2):
            for tower in ("cls_convs", "reg_convs"):
                if depthwise:          # DWConv (yolo_head.py:42): depth-wise 3x3 (weight (C,1,3,3)) + point-wise 1x1
                    conv_bn(f"{tower}.{k}.{i}.dconv", 1, hid, 3)
                    conv_bn(f"{tower}.{k}.{i}.pconv", hid, hid, 1)
                else:
                    conv_bn(f"{tower}.{k}.{i}", hid, hid, 3)
        for name, co in (("cls_preds", num_classes), ("reg_preds", 4), ("obj_preds", 1)):
            p[f"{name}.{k}.weight"] = (torch.rand((co, hid, 1, 1), generator=g) * 2 - 1) / math.sqrt(hid)
            if name == "reg_preds":
                p[f"{name}.{k}.bias"] = (torch.rand(co, generator=g) * 2 - 1) / math.sqrt(hid)
            else:
                p[f"{name}.{k}.bias"] = torch.full((co,), -math.log((1 - prior_prob) / prior_prob))
    return p


# --------------------------------------------------------------------------- init helpers
def mlp_inner_dim(C: int, ratio: int = 4) -> int:
    """ops.py:157."""
    return math.floor(int(C * ratio) * 2 / 3 / 32) * 32


def init_backbone_params(cfg: BackboneCfg, seed: int = 0, ls_init: float = 1e-5, dws_conv: Optional[str] = None, dws_kernel: int = 3) -> Params:
    """random-init parameter dict with the reference's names/shapes.

    The draw order is NOT the reference's (App. D-15); weights always travel by dict, never by seed.
    """
    g = torch.Generator().manual_seed(seed)

    def U(shape, fan_in):
        b = 1.0 / math.sqrt(fan_in)
        return (torch.rand(shape, generator=g) * 2 - 1) * b

    p: Params = {}
    cin = cfg.input_channels
    for s, C in enumerate(cfg.stage_dims):
        f = cfg.patch_size if s == 0 else 2
        k = 2 * (f - 1) + 1
        pre = f"stages.{s}."
        p[pre + "downsample_cf2cl.conv.weight"] = U((C, cin, k, k), cin * k * k)
        p[pre + "downsample_cf2cl.norm.weight"] = torch.ones(C)
        p[pre + "downsample_cf2cl.norm.bias"] = torch.zeros(C)
        inner = mlp_inner_dim(C)
        for bi in range(cfg.num_blocks[s]):
            a = f"{pre}att_blocks.{bi}.att."
            for m in ("win_attn.", "grid_attn."):
                p[a + m + "qkv.weight"] = U((3 * C, C), C)
                p[a + m + "qkv.bias"] = U((3 * C,), C)
                p[a + m + "proj.weight"] = U((C, C), C)
                p[a + m + "proj.bias"] = U((C,), C)
                for nm in ("norm1", "norm2"):
                    p[a + m + nm + ".weight"] = torch.ones(C)
                    p[a + m + nm + ".bias"] = torch.zeros(C)
                p[a + m + "ls1.gamma"] = torch.full((C,), ls_init)
                p[a + m + "ls2.gamma"] = torch.full((C,), ls_init)
                p[a + m + "mlp.net.0.proj.weight"] = U((2 * inner, C), C)
                p[a + m + "mlp.net.0.proj.bias"] = U((2 * inner,), C)
                p[a + m + "mlp.net.2.weight"] = U((C, inner), inner)
                p[a + m + "mlp.net.2.bias"] = U((C,), inner)
            if bi == 0:
                p[a + "to_scores.weight"] = U((C, C), C)
                p[a + "to_scores.bias"] = U((C,), C)
                p[a + "to_controls.weight"] = torch.ones(C, 20)
        p[pre + "lstm.conv1x1.weight"] = U((4 * C, 2 * C, 1, 1), 2 * C)
        p[pre + "lstm.conv1x1.bias"] = U((4 * C,), 2 * C)
        if dws_conv is not None:      # "hidden": dws_conv_only_hidden=True; "xh": on cat(x, h)  (drawn last: the other tensors keep their values)
            cd = C if dws_conv == "hidden" else 2 * C
            p[pre + "lstm.conv3x3_dws.weight"] = U((cd, 1, dws_kernel, dws_kernel), dws_kernel * dws_kernel)
            p[pre + "lstm.conv3x3_dws.bias"] = U((cd,), dws_kernel * dws_kernel)
        cin = C
    return p


def pafpn_conv_list(in_channels=(128, 256, 512), depth: float = 0.67, depthwise: bool = False):
    """(name, cin, cout, ksize, stride) for all conv-BN-SiLU units of YOLOPAFPN (yolo_pafpn.py:50-98).  depthwise: the 3x3 units
    (Bottleneck.conv2, bu_conv*) are DWConvs (network_blocks.py:57-76) = `<name>.dconv` (depth-wise: cin listed as 1, the weight's
    second dimension) + `<name>.pconv` (1x1)."""
    c0, c1, c2 = in_channels
    n = round(3 * depth)
    out = [("lateral_conv0", c2, c1, 1, 1)]

    def conv3(name, ci, co, stride):
        if depthwise:
            return [(f"{name}.dconv", 1, ci, 3, stride), (f"{name}.pconv", ci, co, 1, 1)]
        return [(name, ci, co, 3, stride)]

    def csp(name, ci, co):
        hid = int(co * 0.5)
        l = [(f"{name}.conv1", ci, hid, 1, 1), (f"{name}.conv2", ci, hid, 1, 1)]
        for i in range(n):
            l += [(f"{name}.m.{i}.conv1", hid, hid, 1, 1)] + conv3(f"{name}.m.{i}.conv2", hid, hid, 1)
        l.append((f"{name}.conv3", 2 * hid, co, 1, 1))
        return l

    out += csp("C3_p4", 2 * c1, c1)
    out.append(("reduce_conv1", c1, c0, 1, 1))
    out += csp("C3_p3", 2 * c0, c0)
    out += conv3("bu_conv2", c0, c0, 2)
    out += csp("C3_n3", 2 * c0, c1)
    out += conv3("bu_conv1", c1, c1, 2)
    out += csp("C3_n4", 2 * c1, c2)
    return out


def init_pafpn_params(in_channels=(128, 256, 512), depth: float = 0.67, seed: int = 1, depthwise: bool = False):
    g = torch.Generator().manual_seed(seed)
    p: Params = {}
    for name, ci, co, k, _s in pafpn_conv_list(in_channels, depth, depthwise):
        b = 1.0 / math.sqrt(ci * k * k)
        p[name + ".conv.weight"] = (torch.rand((co, ci, k, k), generator=g) * 2 - 1) * b
        p[name + ".bn.weight"] = torch.ones(co)
        p[name + ".bn.bias"] = torch.zeros(co)
        p[name + ".bn.running_mean"] = torch.zeros(co)
        p[name + ".bn.running_var"] = torch.ones(co)
    return p


# --------------------------------------------------------------------------- (f)2 sequence bookkeeping
def select_backbone_features(feature_seq: Sequence[Dict[int, Tensor]], indices_seq: Sequence[Optional[Sequence[int]]]) -> Optional[Dict[int, Tensor]]:
    """BackboneFeatureSelector (modules/utils/detection.py:24-47) as the training step drives it (modules/detection.py:161-171):
    per timestep the features of the samples that carry labels, v[selected_indices], concatenated over the timesteps.
    indices_seq[t] = None / [] : the timestep contributes nothing (the reference only calls add_backbone_features when
    len(current_labels) > 0)."""
    feats: Dict[int, List[Tensor]] = {}
    for f, idx in zip(feature_seq, indices_seq):
        if idx is None or len(idx) == 0:
            continue
        for k, v in f.items():
            feats.setdefault(k, []).append(v[list(idx)])
    if not feats:
        return None
    return {k: torch.cat(v, dim=0) for k, v in feats.items()}


def rnn_states_reset(states, indices_or_bool=None):
    """RNNStates.recursive_reset (modules/utils/detection.py:96-116) on detached states [(h, c)] per stage: zero the selected samples."""
    out = []
    for h, c in states:
        h, c = h.detach().clone(), c.detach().clone()
        for t in (h, c):
            if indices_or_bool is None:
                t[:] = 0
            else:
                t[indices_or_bool] = 0
        out.append((h, c))
    return out


def sequence_train_step(x_seq: Sequence[Tensor], indices_seq, labels: Tensor, bp: Params, fp: Params, hp: Params, cfg: BackboneCfg,
                        prev_states=None, num_classes: int = 3, strides=(8, 16, 32), depth: float = 0.67, kink_log=None):
    """the model part of Module.training_step (modules/detection.py:139-177): L timesteps through the backbone with the recurrent
    states carried (not detached inside the sequence), the features of the labelled (timestep, sample) pairs gathered, ONE
    PAFPN + head + SimOTA loss call on the batched features.  labels: (sum_t len(indices_seq[t]), max_labels, 5) yolox format in
    gather order.  -> (losses dict, final states, P list per timestep)."""
    states, feats_seq, Ps = prev_states, [], []
    for x in x_seq:
        out, states, P = backbone(x, states, bp, cfg, kink_log=kink_log)
        feats_seq.append(out)
        Ps.append(P)
    sel = select_backbone_features(feats_seq, indices_seq)
    fpn_out = pafpn({k: sel[k] for k in (2, 3, 4)}, fp, depth=depth, training=True)
    losses = yolox_head_train(fpn_out, labels, hp, strides, num_classes=num_classes)
    return losses, states, Ps


def proxy_loss(outs: Sequence[Tensor]) -> Tensor:
    """sum_k mean(out_k^2)."""
    return sum((o.float() ** 2).mean() for o in outs)


def synthetic_events(B: int, hw: Tuple[int, int], seed: int = 0, sparsity: float = 0.0,
                     channels: int = 20) -> Tensor:
    """benchmark.py:58-60 protocol: (rand > sparsity).int(), already-padded size."""
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(B, channels, hw[0], hw[1], generator=g) > sparsity).int()


def count_events(B: int, hw: Tuple[int, int], seed: int = 0, density: float = 0.1,
                 valid_hw: Optional[Tuple[int, int]] = None, channels: int = 20) -> Tensor:
    """dataset-like count-valued input: uint8 counts 1..10 at `density`, zero pad region."""
    g = torch.Generator().manual_seed(seed)
    on = torch.rand(B, channels, hw[0], hw[1], generator=g) < density
    val = torch.randint(1, 11, (B, channels, hw[0], hw[1]), generator=g)
    x = (on * val).to(torch.uint8)
    if valid_hw is not None:
        x[:, :, valid_hw[0]:, :] = 0
        x[:, :, :, valid_hw[1]:] = 0
    return x
