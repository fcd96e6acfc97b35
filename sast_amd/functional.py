"""torch.autograd bindings of the C-ABI ops (include/sast_hip.h).

Layout: activations are fp32 NHWC ("image layout" rows [B*H*W, C]).  Parameter gradients, two modes (`_ParamGrads`):
  * default, IN PLACE: the backward kernels accumulate into `param.grad` (atomicAdd, "+=" semantics, the buffer is created
    zero-filled when missing) and the autograd edge of a parameter returns None -- the same contract as Megatron's fused gradient
    accumulation.  `loss.backward()` fills `.grad` exactly like the reference; the AccumulateGrad node of a parameter still runs
    (with an undefined gradient) AFTER every kernel that writes its `.grad` has been enqueued, so its post hooks -- what
    torch.nn.parallel.DistributedDataParallel reduces from -- see the finished buffer.  `torch.autograd.grad(..., params)` is
    not supported in this mode.
  * AUTOGRAD-VISIBLE (`set_autograd_visible_grads(True)` / SAST_AUTOGRAD_GRADS=1): every backward call accumulates into a fresh
    zero-filled buffer (one allocation + one fill per call for all its parameters) and RETURNS it on the parameter's autograd
    edge, so AccumulateGrad, tensor hooks, `torch.autograd.grad` and anything else that expects ordinary autograd gradients work;
    with `zero_grad(set_to_none=True)` AccumulateGrad adopts the buffer without a copy.  `dist.FlatParams` (TrainStep) needs the
    in-place mode.

There is no CPU implementation: every op raises if its tensors are not on a HIP device.
"""
from __future__ import annotations

import collections
import ctypes as C
import os
from typing import Optional, Tuple

import torch

from . import _lib as L

_DT = {torch.float32: L.DT_F32, torch.int32: L.DT_I32, torch.uint8: L.DT_U8}


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)     # the handle without building a torch.cuda.Stream object per launch


def _stream():
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("sast_amd: tensors must live on a HIP device (no CPU fallback for the SAST hot path)")


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _fill(struct, **kw):
    for k, v in kw.items():
        setattr(struct, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
    return struct


AUTOGRAD_GRADS = os.environ.get("SAST_AUTOGRAD_GRADS", "0") == "1"


def set_autograd_visible_grads(on: bool = True) -> bool:
    """switch between the two parameter-gradient modes of the header (process-wide); returns the previous setting"""
    global AUTOGRAD_GRADS
    prev, AUTOGRAD_GRADS = AUTOGRAD_GRADS, bool(on)
    return prev


def _gbuf(p: Optional[torch.Tensor]):
    """gradient accumulation buffer of a parameter (created zero-filled with the parameter's strides)."""
    if p is None or not p.requires_grad:
        return None
    if p.grad is None:
        p.grad = torch.zeros_like(p)
    g = p.grad
    if g.stride() != p.stride():
        raise RuntimeError("sast_amd: .grad must have the same memory layout as its parameter")
    return g


_SCRATCH_KEEP = collections.deque()
_SCRATCH_KEEP_BYTES = 64 << 20     # a few launches' worth: same-stream reuse by the caching allocator is safe once the launch is enqueued
_SCRATCH_KEEP_MAX = 512            # ... and a bound on the entries: the usual customers are 256 B ... 1 KB vectors
_scratch_bytes = 0


def _scratch_grad(p: torch.Tensor):
    """for frozen parameters (and for the resident zero vectors standing in for absent biases) the kernels still need somewhere to
    accumulate: a throw-away buffer.  It must OUTLIVE the launch it is handed to: callers pass raw pointers, and a buffer freed before
    the launch is handed out again by the caching allocator -- to the next parameter's freshly created `.grad`, which the kernel would
    then corrupt through the stale pointer (found with the attention_bias=False fixture).  Call sites hold their scratch buffers in a
    local until the launch is enqueued (that alone is sufficient: the caching allocator only re-uses the memory on the same stream,
    behind the launch); as a second line of defence the most recent ones are also kept alive here, bounded by BYTES (a model with
    large frozen weights must not pin hundreds of MB for the life of the process) and by COUNT (thousands of tiny vectors would make
    every call walk a long queue and pin as many allocator blocks)."""
    global _scratch_bytes
    t = torch.zeros_like(p)
    _SCRATCH_KEEP.append(t)
    _scratch_bytes += t.numel() * t.element_size()
    while len(_SCRATCH_KEEP) > 1 and (_scratch_bytes > _SCRATCH_KEEP_BYTES or len(_SCRATCH_KEEP) > _SCRATCH_KEEP_MAX):
        old = _SCRATCH_KEEP.popleft()
        _scratch_bytes -= old.numel() * old.element_size()
    return t


def _g(p):
    g = _gbuf(p)
    return g if g is not None else (_scratch_grad(p) if p is not None else None)


def _like_view(flat: torch.Tensor, p: torch.Tensor):
    """`flat` (1-D, p.numel() elements) seen with p's shape AND its exact strides (dense contiguous or channels-last 4-D: the memory
    orders the kernels accept), else None.  as_strided rather than view + permute: for size-1 dimensions several stride tuples describe
    the same memory, and `.grad` is compared with its parameter stride for stride."""
    if p.is_contiguous() or (p.dim() == 4 and p.permute(0, 2, 3, 1).is_contiguous()):
        return flat.as_strided(p.size(), p.stride())
    return None


class _ParamGrads:
    """the parameter gradients of ONE Function.backward call: `pg[i]` is the buffer the kernels accumulate parameter i's gradient into
    (None for a None parameter), `pg.out()` what the call returns on the parameters' autograd edges.
    In-place mode: `p.grad` (created zero-filled) / None.  Autograd-visible mode: views of ONE fresh zero-filled allocation / the same
    views (AccumulateGrad adopts or adds them).  Frozen parameters and non-parameter stand-ins (resident zero biases) get a throw-away
    buffer and return None in both modes.  The object holds every buffer until the caller drops it behind the launch."""

    def __init__(self, *params):
        self.bufs, self.ret = [], []
        if AUTOGRAD_GRADS and torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            # AccumulateGrad adds the returned gradients on the stream each node was CREATED on, not on the capture stream: a hipGraph
            # captured around backward() in this mode replays garbage (measured, round 5).  Any capturing caller is refused, not only
            # TrainStep.capture.
            raise RuntimeError("sast_amd: the autograd-visible gradient mode (set_autograd_visible_grads / SAST_AUTOGRAD_GRADS=1) cannot run "
                               "inside a stream capture: capture the backward in the in-place gradient mode")
        if not AUTOGRAD_GRADS:
            for p in params:
                self.bufs.append(_g(p))
                self.ret.append(None)
            return
        live = [p for p in params if p is not None and p.requires_grad]
        n = sum((p.numel() + 3) // 4 * 4 for p in live)
        flat = torch.zeros(n, device=live[0].device, dtype=torch.float32) if live else None
        off = 0
        for p in params:
            if p is None:
                self.bufs.append(None)
                self.ret.append(None)
            elif not p.requires_grad:
                self.bufs.append(_scratch_grad(p))
                self.ret.append(None)
            else:
                v = _like_view(flat[off:off + p.numel()], p) if p.dtype == torch.float32 else None
                if v is None:
                    v = torch.zeros_like(p)
                off += (p.numel() + 3) // 4 * 4
                self.bufs.append(v)
                self.ret.append(v)

    def __getitem__(self, i):
        return self.bufs[i]

    def out(self):
        return tuple(self.ret)


# ---- deferred weight gradients (include/sast_hip.h: sast_dw_defer ...; csrc/k_defer.hip).  Owned by training.TrainStep: while deferral is
# on, the backward entry points park their weight-gradient jobs in the library's queue and the owner flushes them on a side stream.  The
# parked jobs hold RAW pointers to the upstream gradients, workspaces, saved activations and row counts of the backward call that parked
# them, so every backward that can park hands its locals to `_dw_hold`; the owner releases them (`dw_release`) once the stream it keeps
# allocating on is ordered behind the flushed launches.
_DW_DEFER = False
_DW_HOLD = []


def dw_defer(on: bool, min_rows: int = 0, max_rows: int = 0) -> bool:
    """switch the library's deferral of weight-gradient jobs on / off (process-wide); returns the previous setting.
    min_rows / max_rows: only jobs whose reduction runs over that many rows are parked (0 = unbounded)."""
    global _DW_DEFER
    prev, _DW_DEFER = _DW_DEFER, bool(on)
    L.lib().sast_dw_defer_rows(int(min_rows), int(max_rows))
    L.lib().sast_dw_defer(int(bool(on)))
    return prev


def dw_pending() -> int:
    return int(L.lib().sast_dw_pending())


def dw_flush():
    """enqueue every parked weight-gradient job on the CURRENT stream (the caller has ordered it behind the backward kernels)"""
    L.check(L.lib().sast_dw_flush(_stream()), "dw_flush")


def dw_discard() -> int:
    return int(L.lib().sast_dw_discard())


def dw_release():
    """drop the references held for parked jobs (the caller has ordered its allocating stream behind the flushed launches)"""
    _DW_HOLD.clear()


def _dw_hold(objs):
    if _DW_DEFER:
        _DW_HOLD.append(objs)


def _consume(ctx, what: str):
    """the backward kernels of these ops CONSUME scratch accumulators that the forward kernels cleared (BatchNorm-backward sums,
    the LayerScale / fc2 / proj raw gradients, d(scale) of the STP controls): a second backward over the same forward
    (retain_graph=True, or two losses back-propagated separately) would add into stale sums.  Refuse it loudly."""
    if getattr(ctx, "_sast_consumed", False):
        raise RuntimeError(f"sast_amd: {what}: backward was already run for this forward; a second backward over a retained graph "
                           "is not supported (the fused backward consumes scratch accumulators cleared by the forward kernels). "
                           "Sum the losses and call backward once, or re-run the forward.")
    ctx._sast_consumed = True


def is_channels_last_weight(w: torch.Tensor) -> bool:
    return w.dim() == 4 and w.permute(0, 2, 3, 1).is_contiguous()


# ---------------------------------------------------------------------------------------------- a1
@torch.no_grad()
def non_zero_ratio(x: torch.Tensor, pad_hw=None) -> torch.Tensor:
    """sast_rnn.py:45-60.  x (B,Cin,H,W) NCHW {uint8,int32,float32} -> (B,4,Cin) fp32.
    pad_hw: x stands for its zero padding (bottom / right) to this size (InputPadderFromShape, utils/padding.py:29-53)."""
    _need_gpu(x)
    if x.dtype not in _DT:
        x = x.float()
    x = x.contiguous()
    B, Cin, H, W = x.shape
    r = torch.empty(B, 4, Cin, device=x.device, dtype=torch.float32)
    cnt = torch.empty(B * 4 * Cin, device=x.device, dtype=torch.int32)
    if pad_hw is not None and tuple(pad_hw) != (H, W):
        L.check(L.lib().sast_nzratio_padded(x.data_ptr(), _DT[x.dtype], B, Cin, H, W, int(pad_hw[0]), int(pad_hw[1]), cnt.data_ptr(),
                                            r.data_ptr(), _stream()), "nzratio_padded")
        return r
    L.check(L.lib().sast_nzratio(x.data_ptr(), _DT[x.dtype], B, Cin, H, W, cnt.data_ptr(), r.data_ptr(), _stream()), "nzratio")
    return r


@torch.no_grad()
def nchw_to_nhwc_float(x: torch.Tensor, pad_hw=None) -> torch.Tensor:
    """x.float() + nChw_2_nhwC (sast_rnn.py:153, ops.py:19-24) in one pass; with pad_hw also the zero padding of
    InputPadderFromShape (utils/padding.py:29-53), so a uint8 event tensor is read exactly once, unpadded."""
    _need_gpu(x)
    if x.dtype not in _DT:
        x = x.float()
    x = x.contiguous()
    B, Cc, H, W = x.shape
    if pad_hw is not None and tuple(pad_hw) != (H, W):
        Hp, Wp = int(pad_hw[0]), int(pad_hw[1])
        y = torch.empty(B, Hp, Wp, Cc, device=x.device, dtype=torch.float32)
        L.check(L.lib().sast_nchw_to_nhwc_padded(x.data_ptr(), _DT[x.dtype], B, Cc, H, W, Hp, Wp, y.data_ptr(), _stream()), "nchw_to_nhwc_padded")
        return y
    y = torch.empty(B, H, W, Cc, device=x.device, dtype=torch.float32)
    L.check(L.lib().sast_nchw_to_nhwc(x.data_ptr(), _DT[x.dtype], B, Cc, H, W, y.data_ptr(), _stream()), "nchw_to_nhwc")
    return y


_PREP_WS = {}


STEM_U8 = os.environ.get("SAST_STEM_U8", "1") != "0"     # uint8 event tensors stay bytes up to the stem conv's loaders


@torch.no_grad()
def input_prep(x: torch.Tensor, pad_hw=None, ws_cache: Optional[dict] = None, keep_bytes: bool = False):
    """non_zero_ratio (sast_rnn.py:45-60) + x.float() + zero padding to pad_hw (utils/padding.py:29-53) + NCHW->NHWC (ops.py:19-24)
    in ONE launch that reads the event tensor once -> (r (B,4,C) fp32, x_nhwc (B,Hp,Wp,C) fp32).  Falls back to the two separate
    launches for shapes the fused kernel does not cover (padded sizes that are not multiples of 32, channel counts other than 20).
    keep_bytes (uint8 input, the dataset's storage type): x_nhwc stays uint8 -- only `downsample_ln` (the stem) may consume it; its
    loaders do the `.float()`.  Ignored (fp32 result) for other dtypes and for the fallback shapes."""
    _need_gpu(x)
    if x.dtype not in _DT:
        x = x.float()
    x = x.contiguous()
    B, Cc, H, W = x.shape
    Hp, Wp = (int(pad_hw[0]), int(pad_hw[1])) if pad_hw is not None else (H, W)
    if H % 4 or W % 4 or Hp % 32 or Wp % 32 or Cc != 20 or ((Hp // 32) * (Wp // 32)) % 2 or Hp < H or Wp < W:
        pad = (Hp, Wp) if (Hp, Wp) != (H, W) else None
        return non_zero_ratio(x, pad), nchw_to_nhwc_float(x, pad)
    # scratch (per-sample counters + the ticket of the "last workgroup"): zero on entry, zero on exit -- cleared when it is created,
    # the kernel's last workgroup leaves it clean.  Owned by the caller (`ws_cache`: RNNDetector keeps one per module instance, whose
    # calls are sequential) or, for bare calls, kept per (device, STREAM, B, C): two streams calling concurrently must not share
    # counters.  It must not be born inside a stream capture (the zero fill would be captured and the buffer would belong to the
    # graph's private pool): one un-captured warm-up call first, as torch's graph capture needs anyway.
    cache = _PREP_WS if ws_cache is None else ws_cache
    key = (x.device, B, Cc) if ws_cache is not None else (x.device, torch.cuda.current_stream().cuda_stream, B, Cc)
    ws = cache.get(key)
    if ws is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("sast_amd: input_prep needs one un-captured warm-up call (per model and batch size) before graph capture")
        ws = cache[key] = torch.zeros(B * 4 * Cc + 1, device=x.device, dtype=torch.int32)
    r = torch.empty(B, 4, Cc, device=x.device, dtype=torch.float32)
    u8 = keep_bytes and x.dtype == torch.uint8
    y = torch.empty(B, Hp, Wp, Cc, device=x.device, dtype=torch.uint8 if u8 else torch.float32)
    try:
        if u8:
            L.check(L.lib().sast_input_prep_u8(x.data_ptr(), B, Cc, H, W, Hp, Wp, ws.data_ptr(), r.data_ptr(), y.data_ptr(), _stream()), "input_prep_u8")
        else:
            L.check(L.lib().sast_input_prep(x.data_ptr(), _DT[x.dtype], B, Cc, H, W, Hp, Wp, ws.data_ptr(), r.data_ptr(), y.data_ptr(), _stream()), "input_prep")
    except Exception:
        cache.pop(key, None)         # a failed launch may leave counters / the ticket non-zero: never reuse this buffer
        raise
    return r, y


def as_nhwc(x: torch.Tensor) -> torch.Tensor:
    """(B,C,H,W) logical NCHW -> (B,H,W,C) contiguous; zero-copy when x is already channels-last in memory."""
    v = x.permute(0, 2, 3, 1)
    return v if v.is_contiguous() else v.contiguous()


def as_nchw_view(x_nhwc: torch.Tensor) -> torch.Tensor:
    """(B,H,W,C) contiguous -> logical (B,C,H,W) view (channels_last memory format, no copy)."""
    return x_nhwc.permute(0, 3, 1, 2)


MEAN_SQUARE_BLOCKS = 128   # include/sast_hip.h: SAST_MEAN_SQUARE_BLOCKS


class _MeanSquares(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *xs):
        _need_gpu(*xs)
        xs = tuple(x.contiguous() for x in xs)
        k = len(xs)
        ptrs = (C.c_void_p * k)(*[x.data_ptr() for x in xs])
        ns = (C.c_size_t * k)(*[x.numel() for x in xs])
        partials = torch.empty(k * MEAN_SQUARE_BLOCKS, device=xs[0].device, dtype=torch.float32)
        L.check(L.lib().sast_mean_square_fwd(ptrs, ns, k, partials.data_ptr(), _stream()), "mean_square_fwd")
        ctx.save_for_backward(*xs)
        return partials

    @staticmethod
    def backward(ctx, g):
        xs = ctx.saved_tensors
        k = len(xs)
        g_stride = 1
        if g.stride(0) == 0:       # the expanded scalar gradient of `.sum()`: read element 0 for every partial, no copy
            g, g_stride = g.as_strided((1,), (1,)), 0
        else:
            g = g.contiguous()
        dxs = tuple(torch.empty_like(x) for x in xs)
        ptrs = (C.c_void_p * k)(*[x.data_ptr() for x in xs])
        dptrs = (C.c_void_p * k)(*[d.data_ptr() for d in dxs])
        ns = (C.c_size_t * k)(*[x.numel() for x in xs])
        L.check(L.lib().sast_mean_square_bwd(ptrs, ns, k, g.data_ptr(), g_stride, dptrs, _stream()), "mean_square_bwd")
        return dxs


def mean_squares(*xs: torch.Tensor) -> torch.Tensor:
    """sum_t mean(x_t ** 2) of up to 4 fp32 tensors with one launch forward and one backward (bench.py's synthetic
    objective on the PAFPN outputs; the partial sums are added by torch)."""
    if not 1 <= len(xs) <= 4 or any(x.dtype != torch.float32 or x.numel() % 4 for x in xs):
        raise ValueError("mean_squares: 1..4 fp32 tensors with numel % 4 == 0")
    return _MeanSquares.apply(*xs).sum()


class _AddRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, table):
        _need_gpu(x, table)
        x = x.contiguous()
        y = torch.empty_like(x)
        Cc = x.shape[-1]
        rows = x.numel() // Cc
        L.check(L.lib().sast_add_rows(x.data_ptr(), table.data_ptr(), y.data_ptr(), rows, Cc, table.numel() // Cc, _stream()), "add_rows")
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, None


def add_pos_embedding(x: torch.Tensor, table: torch.Tensor) -> torch.Tensor:
    return _AddRows.apply(x, table)


# ---------------------------------------------------------------------------------------------- a2
class _DownsampleLN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, ln_w, ln_b, pe, factor):
        _need_gpu(x, w)
        x = x.contiguous()
        if not is_channels_last_weight(w):
            raise RuntimeError("sast_amd: conv weights must be stored channels_last ([Cout][KH][KW][Cin])")
        B, H, W, Cin = x.shape
        Cout = w.shape[0]
        k = w.shape[-1]
        if k not in (factor, 2 * factor - 1) or w.shape[-2] != k:     # ops.py:70-76: overlap True -> 2f-1 (replicate pad f-1), False -> f (no pad)
            raise RuntimeError(f"sast_amd: a factor-{factor} downsampling conv has a {2 * factor - 1}x{2 * factor - 1} (overlap) or "
                               f"{factor}x{factor} (no overlap) kernel, not {tuple(w.shape[-2:])}")
        no_overlap = int(k == factor)
        Ho, Wo = H // factor, W // factor
        M = B * Ho * Wo
        dev = x.device
        conv_out = torch.empty(M, Cout, device=dev)
        stats = torch.empty(2, M, device=dev)
        y = torch.empty(B, Ho, Wo, Cout, device=dev)
        if x.dtype not in (torch.float32, torch.uint8):
            raise RuntimeError("sast_amd: downsample_ln reads fp32 rows, or (the stem) the uint8 event tensor from input_prep(keep_bytes=True)")
        xdt = _DT[x.dtype]
        a = _fill(L.SastDownArgs(), B=B, H=H, W=W, Cin=Cin, Cout=Cout, factor=factor, x=x, w=w, ln_w=ln_w, ln_b=ln_b,
                  pe=_ptr(pe), conv_out=conv_out, mean=stats[0], rstd=stats[1], y=y, x_dtype=xdt, no_overlap=no_overlap)
        L.check(L.lib().sast_downsample_ln_fwd(C.byref(a), _stream()), "downsample_ln_fwd")
        ctx.save_for_backward(x, conv_out, stats)
        ctx.params = (w, ln_w, ln_b)
        ctx.meta = (B, H, W, Cin, Cout, factor, xdt, no_overlap)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, conv_out, stats = ctx.saved_tensors
        w, ln_w, ln_b = ctx.params
        B, H, W, Cin, Cout, factor, xdt, no_overlap = ctx.meta
        dy = dy.contiguous()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        ws = torch.empty(conv_out.numel(), device=x.device)
        pg = _ParamGrads(w, ln_w, ln_b)
        a = _fill(L.SastDownArgs(), B=B, H=H, W=W, Cin=Cin, Cout=Cout, factor=factor, x=x, w=w, ln_w=ln_w, ln_b=ln_b,
                  conv_out=conv_out, mean=stats[0], rstd=stats[1], dy=dy, dx=_ptr(dx), dw=pg[0], d_ln_w=pg[1], d_ln_b=pg[2], ws=ws,
                  x_dtype=xdt, no_overlap=no_overlap)
        L.check(L.lib().sast_downsample_ln_bwd(C.byref(a), _stream()), "downsample_ln_bwd")
        _dw_hold((x, conv_out, stats, dy, ws, pg))
        return (dx,) + pg.out() + (None, None)


def downsample_ln(x_nhwc, w, ln_w, ln_b, pe, factor):
    return _DownsampleLN.apply(x_nhwc, w, ln_w, ln_b, pe, factor)


class _MaskToken(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mask, token, pe):
        _need_gpu(x, mask, token)
        y = x.contiguous().clone()
        Cc = y.shape[-1]
        rows = y.numel() // Cc
        m = mask.to(torch.uint8).contiguous()
        Lt = pe.numel() // Cc if pe is not None else 1
        L.check(L.lib().sast_mask_token_fwd(y.data_ptr(), m.data_ptr(), token.data_ptr(), _ptr(pe), rows, Cc, Lt, _stream()), "mask_token_fwd")
        ctx.save_for_backward(m)
        ctx.token = token
        return y

    @staticmethod
    def backward(ctx, dy):
        (m,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        Cc = dy.shape[-1]
        pg = _ParamGrads(ctx.token)
        L.check(L.lib().sast_mask_token_bwd(dy.data_ptr(), m.data_ptr(), dx.data_ptr(), pg[0].data_ptr(), dy.numel() // Cc, Cc, _stream()),
                "mask_token_bwd")
        return dx, None, pg.out()[0], None


def mask_token(x_nhwc, token_mask, token, pos_emb_table=None):
    """sast_rnn.py:271-273 on rows that already carry the position embedding: x[token_mask] = mask_token (+ pos_emb)."""
    return _MaskToken.apply(x_nhwc, token_mask, token, pos_emb_table)


# ---------------------------------------------------------------------------------------------- a5
class _ScoreSTP(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xp, r, ws_w, ws_b, wc, amp):
        _need_gpu(xp, r, ws_w)
        ctx.set_materialize_grads(False)   # no zero-filled gradient for the non-differentiable token scores
        xp = xp.contiguous()
        B = xp.shape[0]
        Cc = xp.shape[-1]
        Lt = xp.numel() // (B * Cc)
        assert r.shape[-1] == 20 and r.stride(-1) == 1
        dev = xp.device
        scale = torch.empty(B, Cc, device=dev)
        s = torch.empty(B * Lt, Cc, device=dev)
        xw = torch.empty_like(xp)
        tok = torch.empty(B, Lt, device=dev)
        dscale = torch.empty(B, Cc, device=dev)    # cleared by the forward kernels, accumulated into by the backward
        a = _fill(L.SastScoreArgs(), B=B, L=Lt, C=Cc, r_stride=r.stride(0), amp=amp, xp=xp, r=r, ws_w=ws_w, ws_b=ws_b, wc=wc,
                  scale=scale, s=s, xw=xw, tok=tok, dscale_ws=dscale)
        L.check(L.lib().sast_score_stp_fwd(C.byref(a), _stream()), "score_stp_fwd")
        ctx.save_for_backward(xp, r, scale, s, dscale)
        ctx.params = (ws_w, ws_b, wc)
        ctx.meta = (B, Lt, Cc, amp)
        ctx.mark_non_differentiable(tok)
        return xw, tok

    @staticmethod
    def backward(ctx, dxw, _dtok):
        xp, r, scale, s, dscale = ctx.saved_tensors
        ws_w, ws_b, wc = ctx.params
        B, Lt, Cc, amp = ctx.meta
        if dxw is None:
            return None, None, None, None, None, None
        _consume(ctx, "score_stp")
        dxw = dxw.contiguous()
        dxp = torch.empty_like(xp)
        ws = torch.empty(B * Lt * Cc + B * Cc, device=xp.device)
        pg = _ParamGrads(ws_w, ws_b, wc)
        a = _fill(L.SastScoreArgs(), B=B, L=Lt, C=Cc, r_stride=r.stride(0), amp=amp, xp=xp, r=r, ws_w=ws_w, ws_b=ws_b, wc=wc,
                  scale=scale, s=s, dxw=dxw, dxp=dxp, d_ws_w=pg[0], d_ws_b=pg[1], d_wc=pg[2], ws=ws, dscale_ws=dscale)
        L.check(L.lib().sast_score_stp_bwd(C.byref(a), _stream()), "score_stp_bwd")
        _dw_hold((xp, r, scale, s, dscale, dxw, ws, pg))
        return (dxp, None) + pg.out() + (None,)


def score_stp(xp, r, ws_w, ws_b, wc, amp) -> Tuple[torch.Tensor, torch.Tensor]:
    return _ScoreSTP.apply(xp, r, ws_w, ws_b, wc, float(amp))


# ---------------------------------------------------------------------------------------------- a6-a8
def mask_words(T: int) -> int:
    """64-bit words of a group's kept-token mask (SastSel.mask): 2 for partitions of up to 128 tokens, 4 up to 256"""
    if T > 256:
        raise NotImplementedError("sast_amd: partitions of more than 256 tokens are not supported by the selection kernels")
    return 2 if T <= 128 else 4


class Selection:
    """device-resident result of the window/token selection (SastSel).  No host sync unless a
    reference-style index list is requested."""

    def __init__(self, B, H, W, ph, pw, mode, device):
        self.B, self.H, self.W, self.ph, self.pw, self.mode = B, H, W, ph, pw, mode
        self.T = ph * pw
        self.N = (H // ph) * (W // pw)
        self.L = H * W
        nw = B * self.N
        buf = torch.empty(5 * nw + 4 + 3 * B * self.L, device=device, dtype=torch.int32)
        self.win_keep, self.K, self.row_off, self.win_rank, self.pack_rows = (buf[i * nw:(i + 1) * nw] for i in range(5))
        o = 5 * nw
        self.counts = buf[o:o + 4]
        self.tok_slot = buf[o + 4:o + 4 + B * self.L]
        self.row_tok = buf[o + 4 + B * self.L:o + 4 + 2 * B * self.L]
        self.row_seg = buf[o + 4 + 2 * B * self.L:]
        self.mask = torch.empty(nw, mask_words(self.T), device=device, dtype=torch.int64)
        self._buf = buf
        self.tok = None  # scores the selection was computed from (for index-list export)

    def fill_struct(self, s):
        _fill(s, win_keep=self.win_keep, mask=self.mask, K=self.K, row_off=self.row_off, win_rank=self.win_rank,
              counts=self.counts, tok_slot=self.tok_slot, row_tok=self.row_tok, pack_rows=self.pack_rows, row_seg=self.row_seg)
        return s

    def build_packs(self):
        """attention packs of a selection that was NOT produced by sast_select (index-list constructor below)"""
        if not hasattr(self, "pack_rows"):
            self.pack_rows = torch.zeros(self.B * self.N, device=self.K.device, dtype=torch.int32)
            self.row_seg = torch.zeros(self.B * self.L, device=self.K.device, dtype=torch.int32)
        L.check(L.lib().sast_select_packs(C.byref(self.struct()), self.B * self.N, self.T, _stream()), "select_packs")

    def struct(self):
        return self.fill_struct(L.SastSel())

    # ---- host-side views (synchronising; parity tests and the reference-style index_list only)
    def num_kept_tokens(self) -> int:
        return int(self.counts[0].item())

    def index_window(self) -> torch.Tensor:
        """ascending flat ids of kept windows (get_score_index_2d21d, SAST.py:258-267)."""
        iw = torch.nonzero(self.win_keep).view(-1)
        return iw

    def K_list(self) -> torch.Tensor:
        return self.K[self.win_keep.bool()].long()

    def asy_index(self) -> torch.Tensor:
        """ascending ids m*T+t of kept tokens in the compacted-window space (SAST.py:279-281)."""
        bits = self._mask_bits()[self.win_keep.bool()]
        return torch.nonzero(bits.reshape(-1)).view(-1)

    def _mask_bits(self) -> torch.Tensor:
        ar = torch.arange(64, device=self.mask.device, dtype=torch.int64)
        words = [(self.mask[:, k:k + 1] >> ar) & 1 for k in range(self.mask.shape[1])]
        return torch.cat(words, dim=1)[:, : self.T].bool()

    def group_token_ids(self) -> torch.Tensor:
        """(N, T) token index l = y*W + x of slot t of group n (ops.py:189-220 maps)."""
        H, W, ph, pw = self.H, self.W, self.ph, self.pw
        idx = torch.arange(H * W).view(1, H, W, 1)
        if self.mode == 0:
            g = idx.view(1, H // ph, ph, W // pw, pw, 1).permute(0, 1, 3, 2, 4, 5)
        else:
            g = idx.view(1, ph, H // ph, pw, W // pw, 1).permute(0, 2, 4, 1, 3, 5)
        return g.reshape(self.N, self.T)

    def to_index_list(self):
        """[index_window, index_token, padding_index, asy_index, K] as the reference builds them
        (SAST.py:120-123).  index_token / padding_index carry the top-k fillers, whose order is
        unspecified in the reference (topk sorted=False); they are reproduced from the token scores."""
        iw, asy, K = self.index_window(), self.asy_index(), self.K_list()
        if self.tok is None:
            return [iw, None, None, asy, K]
        gid = self.group_token_ids().to(self.tok.device)
        tokg = self.tok.view(self.B, self.L)[:, gid].reshape(self.B * self.N, self.T)
        nt = tokg[iw].softmax(-1)
        kmax = int(K.max().item()) if K.numel() else 0
        top = torch.topk(nt, k=kmax, dim=1, largest=True, sorted=False)[1]
        base = torch.arange(0, nt.shape[0] * self.T, self.T, device=nt.device).view(-1, 1)
        it = (top + base).view(-1)
        pad = it[torch.isin(it, asy, assume_unique=True, invert=True)]
        return [iw, it, pad, asy, K]

    def __iter__(self):
        return iter(self.to_index_list())

    def __getitem__(self, i):
        return self.to_index_list()[i]

    def __len__(self):
        return 5


@torch.no_grad()
def select(tok: torch.Tensor, B, H, W, ph, pw, mode, bounce) -> Selection:
    _need_gpu(tok)
    if H % ph or W % pw:
        raise AssertionError(f"map {H}x{W} must be divisible by partition ({ph},{pw})")  # ops.py:191-192
    sel = Selection(B, H, W, ph, pw, mode, tok.device)
    tok = tok.contiguous()
    s = sel.struct()
    L.check(L.lib().sast_select(tok.data_ptr(), B, H, W, ph, pw, mode, float(bounce), C.byref(s), _stream()), "select")
    sel.tok = tok
    return sel


@torch.no_grad()
def select_pair(tok: torch.Tensor, B, H, W, ph, pw, bounce):
    """window-layer and grid-layer selection of one SAST block (SAST.py:120-123 and :141-147 use the same token scores) in
    the same four launches -> (Selection mode 0, Selection mode 1), identical to two `select` calls."""
    _need_gpu(tok)
    if H % ph or W % pw:
        raise AssertionError(f"map {H}x{W} must be divisible by partition ({ph},{pw})")  # ops.py:191-192
    s1, s2 = Selection(B, H, W, ph, pw, 0, tok.device), Selection(B, H, W, ph, pw, 1, tok.device)
    tok = tok.contiguous()
    a, b = s1.struct(), s2.struct()
    L.check(L.lib().sast_select_pair(tok.data_ptr(), B, H, W, ph, pw, float(bounce), C.byref(a), C.byref(b), _stream()), "select_pair")
    s1.tok = s2.tok = tok
    return s1, s2


def selection_from_index_lists(index_window, asy_index, K, n_groups: int, T: int, device) -> Selection:
    """build the device-side selection from reference-style lists in PARTITIONED layout
    (token id = group*T + slot): used by the stand-alone MS_WSA.forward drop-in."""
    sel = Selection.__new__(Selection)
    sel.B, sel.H, sel.W, sel.ph, sel.pw, sel.mode = 1, n_groups * T, 1, T, 1, 0
    sel.T, sel.N, sel.L = T, n_groups, n_groups * T
    iw = index_window.to(device).long()
    asy = asy_index.to(device).long()
    Kl = K.to(device).long()
    i32 = dict(device=device, dtype=torch.int32)
    sel.win_keep = torch.zeros(n_groups, **i32)
    sel.win_keep[iw] = 1
    sel.K = torch.zeros(n_groups, **i32)
    sel.K[iw] = Kl.int()
    # exclusive prefix of K over ALL groups (a dropped group gets the offset of the next kept row, as on the device: a pack may be
    # led by a dropped group)
    sel.row_off = (torch.cumsum(sel.K.long(), 0) - sel.K.long()).int()
    sel.win_rank = torch.full((n_groups,), -1, **i32)
    sel.win_rank[iw] = torch.arange(len(iw), **i32)
    total = int(asy.numel())
    sel.counts = torch.tensor([total, len(iw), total, 0], **i32)
    tokens = iw[asy // T] * T + asy % T          # partitioned-layout token of every compact row
    sel.tok_slot = torch.full((n_groups * T,), -1, **i32)
    sel.tok_slot[tokens] = torch.arange(total, **i32)
    sel.row_tok = torch.zeros(n_groups * T, **i32)
    sel.row_tok[:total] = tokens.int()
    # kept-token bitmask of every group (two 64-bit words): the kernels test it to tell kept from passed-through tokens
    nwords = mask_words(T)
    bits = torch.zeros(n_groups, 64 * nwords, device=device, dtype=torch.int64)
    bits[iw[asy // T], asy % T] = 1
    sh = torch.arange(64, device=device, dtype=torch.int64)
    # bit 63 wraps into the sign: the same two's-complement word the device writes
    sel.mask = torch.stack([(bits[:, 64 * k:64 * (k + 1)] << sh).sum(1) for k in range(nwords)], dim=1).contiguous()
    sel.tok = None
    sel.build_packs()
    return sel


class DeviceCount:
    """sum of 0-dim device integers that is only evaluated when somebody looks at it (int(), .item(), .tensor()).
    The reference returns python ints for the kept-token count P (SAST.py:136,159; >= 8 host syncs per block); on the
    device-resident path even a 2-element integer add is a ~5 us kernel launch per block, so the additions are deferred."""
    __slots__ = ("terms",)

    def __init__(self, terms=()):
        self.terms = tuple(terms)

    def __add__(self, other):
        if isinstance(other, DeviceCount):
            return DeviceCount(self.terms + other.terms)
        if isinstance(other, int) and other == 0:
            return self
        return DeviceCount(self.terms + (other,))

    __radd__ = __add__

    def tensor(self) -> torch.Tensor:
        ts = [t if isinstance(t, torch.Tensor) else torch.as_tensor(t) for t in self.terms]
        return torch.stack([t.to(ts[0].device).long() for t in ts]).sum() if ts else torch.zeros((), dtype=torch.long)

    def item(self) -> int:
        return int(self.tensor().item())

    __int__ = item
    __index__ = item

    def __repr__(self):
        return f"DeviceCount({len(self.terms)} terms)"


# ---------------------------------------------------------------------------------------------- a9
_MSWSA_PARAMS = ("ln1_w", "ln1_b", "ln2_w", "ln2_b", "qkv_w", "qkv_b", "proj_w", "proj_b", "ls1",
                 "fc1_w", "fc1_b", "fc2_w", "fc2_b", "ls2", "act_w")      # act_w: the slope of mlp_activation prelu (fp32[1]), None otherwise


_FUSED_ENABLE = True  # tools / tests switch the fused form off to compare the two forms of the layer in one process
# rows (B * L tokens of the layer) from which the one-kernel forward is used: a wave of it runs the whole layer for 32 tokens (~50 us of
# dependent work), which only pays once the rows give every SIMD of the chip about two such waves.  Measured: 1Mpx B = 4 (61 440 rows)
# -1.5 % of the training step, -4.2 % forward only; Gen1 B = 4 (20 480 rows) +1.9 % forward only (profiles/r04_k).  Tests set 0.
FUSED_MIN_ROWS_DEFAULT = int(os.environ.get("SAST_MSWSA_FUSED_MIN_ROWS", "49152"))
_FUSED_MIN_ROWS = FUSED_MIN_ROWS_DEFAULT
MSWSA_FORM_CALLS = {"fused": 0, "chain": 0}     # forward calls per form of the layer (tests assert which one the policy chose)


class _MSWSA(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xin, sel: Selection, eps, cb_tps, dim_head, fused, mlp_act, drop, grad_on, *params):
        _need_gpu(xin)
        xin = xin.contiguous()
        p = dict(zip(_MSWSA_PARAMS, params))
        Cc = xin.shape[-1]
        R = xin.numel() // Cc
        inner = p["fc2_w"].shape[1]
        if Cc % dim_head:
            raise RuntimeError(f"sast_amd: dim ({Cc}) must be a multiple of dim_head ({dim_head})")
        heads = Cc // dim_head
        dev = xin.device
        out = torch.empty_like(xin)
        a = L.SastMswsaArgs()
        _fill(a, B=sel.B, H=sel.H, W=sel.W, C=Cc, ph=sel.ph, pw=sel.pw, mode=sel.mode, inner=inner, eps=eps, dim_head=dim_head, mlp_act=mlp_act, xin=xin, out=out)
        sel.fill_struct(a.sel)
        _fill(a, **{k: _ptr(v) for k, v in p.items()})
        drop_mlp = None
        if drop is not None:       # (d1, d2, mlp mask): DropPath row factors of the two residual branches, nn.Dropout mask of the MLP hidden
            d1, d2, drop_mlp = (tuple(drop) + (None,))[:3]
            drop = None
            if d1 is not None:
                drop = (d1.contiguous(), d2.contiguous())
                if any(d.dtype != torch.float32 or d.numel() < R for d in drop):
                    raise RuntimeError("sast_amd: the DropPath factors must be fp32 with one entry per row upper bound")
                _fill(a, drop1=drop[0], drop2=drop[1])
            if drop_mlp is not None:
                drop_mlp = drop_mlp.contiguous()
                if drop_mlp.dtype != torch.float32 or drop_mlp.numel() < R * inner:
                    raise RuntimeError("sast_amd: the MLP dropout mask must be fp32 [rows upper bound, inner]")
                _fill(a, drop_mlp=drop_mlp)
        # the layer's forward as ONE kernel (csrc/k_mswsa_fused.hip) where the library has that form for the shape; the scratch holds the
        # bf16x3 weight planes the kernel streams.  In training the same kernel also writes the activations the backward reads.
        if (mlp_act == GLU_ACTIVATIONS["prelu"]) != (p["act_w"] is not None):
            raise RuntimeError("sast_amd: mlp_activation prelu needs its slope parameter (act_w), every other activation none")
        fused_floats = L.lib().sast_mswsa_fused_ws_floats(Cc, inner, sel.ph * sel.pw, dim_head, cb_tps) if (_FUSED_ENABLE and fused and mlp_act == 0 and drop is None and drop_mlp is None and R >= _FUSED_MIN_ROWS) else 0
        # a backward can follow: grad mode was on at the call (it is always off inside Function.forward, and `needs_input_grad` says
        # True for trainable parameters under torch.no_grad() too -- `mswsa` samples the mode) and some input wants a gradient
        needs_bwd = bool(grad_on) and any(ctx.needs_input_grad)
        fws = None
        MSWSA_FORM_CALLS["fused" if fused_floats else "chain"] += 1
        if fused_floats:
            fws = torch.empty(fused_floats, device=dev)
            _fill(a, fused_ws=fws)
            if not needs_bwd:       # inference: the layer writes nothing but its output
                L.check(L.lib().sast_mswsa_fwd(C.byref(a), _stream()), "mswsa_fwd (fused)")
                return out
        stats = torch.empty(4, R, device=dev)
        big = torch.empty(R, Cc * 6 + 3 * inner + heads, device=dev)  # one allocation for all saved activations
        # carve [R, width] blocks out of `big` as separate contiguous buffers
        flat = big.view(-1)
        off = 0

        def carve(width):
            nonlocal off
            t = flat[off:off + R * width]
            off += R * width
            return t

        S, QKV, O, Y, UG, Hh, lse = carve(Cc), carve(3 * Cc), carve(Cc), carve(Cc), carve(2 * inner), carve(inner), carve(heads)
        raw = torch.empty(L.lib().sast_mswsa_raw_ws_floats(Cc, inner), device=dev)   # cleared by the forward, used by the backward
        _fill(a, raw_ws=raw, mean1=stats[0], rstd1=stats[1], mean2=stats[2], rstd2=stats[3], S=S, QKV=QKV, O=O, lse=lse, Y=Y, UG=UG, Hh=Hh)
        if cb_tps:
            if R % cb_tps:
                raise RuntimeError(f"sast_amd: Context Broadcasting needs rows ({R}) divisible by tokens per sample ({cb_tps})")
            cb_m, cb_sum = torch.empty(R, Cc, device=dev), torch.empty(R // cb_tps, Cc, device=dev)
            _fill(a, cb_tps=cb_tps, cb_m=cb_m, cb_sum=cb_sum)
        L.check(L.lib().sast_mswsa_fwd(C.byref(a), _stream()), "mswsa_fwd")
        ctx.save_for_backward(xin, stats, big, raw, fws)      # fws: the weight planes the fused kernels of this call pair stream (or None)
        ctx.sel, ctx.params, ctx.eps, ctx.inner, ctx.cb_tps, ctx.dim_head, ctx.mlp_act = sel, params, eps, inner, cb_tps, dim_head, mlp_act
        ctx.drop, ctx.drop_mlp = drop, drop_mlp
        return out

    @staticmethod
    def backward(ctx, dout):
        xin, stats, big, raw, fws = ctx.saved_tensors
        _consume(ctx, "mswsa")
        sel, params, inner = ctx.sel, ctx.params, ctx.inner
        p = dict(zip(_MSWSA_PARAMS, params))
        Cc = xin.shape[-1]
        R = xin.numel() // Cc
        heads = Cc // ctx.dim_head
        dout = dout.contiguous()
        dxin = torch.empty_like(xin)
        flat = big.view(-1)
        off = 0

        def carve(width):
            nonlocal off
            t = flat[off:off + R * width]
            off += R * width
            return t

        S, QKV, O, Y, UG, Hh, lse = carve(Cc), carve(3 * Cc), carve(Cc), carve(Cc), carve(2 * inner), carve(inner), carve(heads)
        ws = torch.empty(L.lib().sast_mswsa_bwd_ws_floats(R, Cc, inner), device=xin.device)
        a = L.SastMswsaArgs()
        _fill(a, B=sel.B, H=sel.H, W=sel.W, C=Cc, ph=sel.ph, pw=sel.pw, mode=sel.mode, inner=inner, eps=ctx.eps, dim_head=ctx.dim_head, mlp_act=ctx.mlp_act, xin=xin,
              mean1=stats[0], rstd1=stats[1], mean2=stats[2], rstd2=stats[3], S=S, QKV=QKV, O=O, lse=lse, Y=Y, UG=UG, Hh=Hh,
              dout=dout, dxin=dxin, ws=ws, raw_ws=raw)
        if fws is not None:
            _fill(a, fused_ws=fws)
        if ctx.drop is not None:
            drop_ws = torch.empty(2 * R * Cc, device=xin.device)
            _fill(a, drop1=ctx.drop[0], drop2=ctx.drop[1], drop_ws=drop_ws)
        if ctx.drop_mlp is not None:
            _fill(a, drop_mlp=ctx.drop_mlp)
        if ctx.cb_tps:
            cb_m, cb_sum = torch.empty(R, Cc, device=xin.device), torch.empty(R // ctx.cb_tps, Cc, device=xin.device)
            _fill(a, cb_tps=ctx.cb_tps, cb_m=cb_m, cb_sum=cb_sum)
        sel.fill_struct(a.sel)
        _fill(a, **{k: _ptr(v) for k, v in p.items()})
        pg = _ParamGrads(*params)                          # held until the launch is enqueued (scratch buffers among them)
        _fill(a, **{"d_" + k: _ptr(pg[i]) for i, k in enumerate(_MSWSA_PARAMS)})
        L.check(L.lib().sast_mswsa_bwd(C.byref(a), _stream()), "mswsa_bwd")
        _dw_hold((xin, stats, big, raw, fws, dout, ws, pg, sel, ctx.drop, ctx.drop_mlp, locals().get("drop_ws"), locals().get("cb_m"), locals().get("cb_sum")))
        return (dxin, None, None, None, None, None, None, None, None) + pg.out()


# include/sast_hip.h: SastMswsaArgs.mlp_act.  Every name of the reference's get_act_layer (layers/create_act.py:62-79, with the defaults of
# the torch modules it maps to); `prelu` carries ONE learnable slope (layers/activations.py:124-131: `...mlp.net.0.act_layer.weight` in the
# reference's state_dict), handed to the kernels as `act_w`.
GLU_ACTIVATIONS = {"gelu": 0, "relu": 1, "silu": 2, "swish": 2, "sigmoid": 3, "tanh": 4, "mish": 5, "relu6": 6, "leaky_relu": 7, "elu": 8,
                   "celu": 8, "selu": 9, "hard_sigmoid": 10, "hardsigmoid": 10, "hard_swish": 11, "hardswish": 11, "hard_mish": 12, "prelu": 13}


def mswsa(xin, sel: Selection, eps: float, params: dict, cb_tokens_per_sample: int = 0, dim_head: int = 32, fused: bool = True,
          mlp_activation: str = "gelu", drop_path=None) -> torch.Tensor:
    """params: dict with the keys of _MSWSA_PARAMS (ls1/ls2 may be None = LayerScale disabled).
    cb_tokens_per_sample > 0 enables Context Broadcasting (SAST.py:240-246) with that many tokens per sample.
    dim_head: 32 or 24 (the widths the reference ships).
    fused: allow the one-kernel forward (csrc/k_mswsa_fused.hip) where the library has it for the shape.  Same results either way; the
    fused form is faster when most tokens are kept (-1.5 % of the dense 1Mpx step) and slower when few are (+1.5 % at 15 % kept: a
    wave runs the whole layer for its <= 32 tokens, a latency the compacted GEMM chain does not have) -- SAST_block passes its AMP.
    mlp_activation: the gate activation of the GLU-MLP (attention_cfg.mlp_activation, layers/create_act.py:62-79): one of
    GLU_ACTIVATIONS; the one-kernel forward exists for "gelu".
    drop_path: None, or (d1, d2[, mlp_mask]) -- d1 / d2 (both or None): fp32 vectors with one entry per row (upper bound B*L; entry m
    belongs to the m-th KEPT row in asy_index order): keep / keep_prob of timm's DropPath on the attention and on the MLP branch
    (SAST.py:188,193,232,248); mlp_mask (or None): fp32 [rows, inner], keep / (1 - p) of the MLP's nn.Dropout (`drop_mlp`, ops.py:167)."""
    if mlp_activation not in GLU_ACTIVATIONS:
        raise NotImplementedError(f"sast_amd: mlp_activation {mlp_activation!r}: the GLU epilogues implement {sorted(GLU_ACTIVATIONS)}")
    return _MSWSA.apply(xin, sel, float(eps), int(cb_tokens_per_sample), int(dim_head), bool(fused), GLU_ACTIVATIONS[mlp_activation],
                        drop_path, torch.is_grad_enabled(), *[params.get(k) if k == "act_w" else params[k] for k in _MSWSA_PARAMS])


# ---------------------------------------------------------------------------------------------- a12
class _LSTM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, h0, c0, w, b, drop=None):
        _need_gpu(x, w)
        ctx.set_materialize_grads(False)
        x = x.contiguous()
        h0 = h0.contiguous() if h0 is not None else None
        c0 = c0.contiguous() if c0 is not None else None
        Cc = x.shape[-1]
        B = x.shape[0]
        Lt = x.numel() // (B * Cc)
        h1, c1 = torch.empty_like(x), torch.empty_like(x)
        gates = torch.empty(B * Lt, 4 * Cc, device=x.device)
        if drop is not None:
            if drop.shape != x.shape or drop.dtype != torch.float32:
                raise RuntimeError("sast_amd: the cell-update dropout mask must be fp32 of the input's NHWC shape")
            drop = drop.contiguous()
        a = _fill(L.SastLstmArgs(), B=B, L=Lt, C=Cc, x=x, h0=_ptr(h0), c0=_ptr(c0), w=w, b=b, h1=h1, c1=c1, gates=gates, drop=_ptr(drop))
        L.check(L.lib().sast_lstm_fwd(C.byref(a), _stream()), "lstm_fwd")
        ctx.save_for_backward(x, h0, c0, c1, gates, drop)
        ctx.params = (w, b)
        ctx.meta = (B, Lt, Cc)
        # h1 twice: a stage's output feeds the next stage AND the FPN / the next time step; handing the two consumers two
        # aliases makes autograd deliver their gradients separately, and the backward kernel adds them while it reads them
        return h1, h1.view_as(h1), c1

    @staticmethod
    def backward(ctx, dh1, dh1b, dc1):
        x, h0, c0, c1, gates, drop = ctx.saved_tensors
        w, b = ctx.params
        B, Lt, Cc = ctx.meta
        if dh1 is None:
            dh1, dh1b = dh1b, None
        if dh1 is None:
            dh1 = torch.zeros_like(x)
        dh1 = dh1.contiguous()
        dh1b = dh1b.contiguous() if dh1b is not None else None
        dc1 = dc1.contiguous() if dc1 is not None else None
        dx = torch.empty_like(x)
        need_h = h0 is not None and ctx.needs_input_grad[1]
        need_c = c0 is not None and ctx.needs_input_grad[2]
        dh0 = torch.empty_like(x) if need_h else None
        dc0 = torch.empty_like(x) if need_c else None
        ws = torch.empty(B * Lt * 4 * Cc, device=x.device)
        pg = _ParamGrads(w, b)
        a = _fill(L.SastLstmArgs(), B=B, L=Lt, C=Cc, x=x, h0=_ptr(h0), c0=_ptr(c0), w=w, b=b, c1=c1, gates=gates, dh1=dh1,
                  dc1=_ptr(dc1), dx=dx, dh0=_ptr(dh0), dc0=_ptr(dc0), dw=pg[0], db=pg[1], ws=ws, dh1b=_ptr(dh1b), drop=_ptr(drop))
        L.check(L.lib().sast_lstm_bwd(C.byref(a), _stream()), "lstm_bwd")
        _dw_hold((x, h0, c0, c1, gates, drop, dh1, dh1b, dc1, ws, pg))
        return (dx, dh0, dc0) + pg.out() + (None,)


def conv_lstm(x_nhwc, h0, c0, w, b, two_h=False, drop_mask=None):
    """-> (h1, c1), or with two_h (h1, h1_alias, c1): two handles on the same h1 for its two consumers (their gradients
    are then summed inside the backward kernel instead of by an autograd add launch).
    drop_mask: fp32 NHWC tensor of x's shape, keep mask / (1 - p) of `cell_update_dropout` (rnn.py:64), None = no dropout"""
    h1, h1b, c1 = _LSTM.apply(x_nhwc, h0, c0, w, b, drop_mask)
    return (h1, h1b if TWO_OUT else h1, c1) if two_h else (h1, c1)


class _DwConv(torch.autograd.Function):
    """depth-wise k x k conv with bias on NHWC rows (DWSConvLSTM2d.conv3x3_dws, rnn.py:24-28).  c0: the conv uses the channels
    [c0, c0 + C) of the parameters (the two halves of a depth-wise conv over cat(x, h) run as two calls on the same parameters)."""

    @staticmethod
    def forward(ctx, x, w, b, c0):
        _need_gpu(x, w)
        x = x.contiguous()
        B, H, W, Cc = x.shape
        k = w.shape[-1]
        if w.dim() != 4 or w.shape[1] != 1 or w.shape[2] != k or not w.is_contiguous() or c0 < 0 or c0 + Cc > w.shape[0]:
            raise RuntimeError("sast_amd: dwconv needs the contiguous Conv2d(groups=C) weight [C, 1, k, k]")
        y = torch.empty_like(x)
        L.check(L.lib().sast_dwconv_fwd(x.data_ptr(), w.data_ptr() + 4 * c0 * k * k, (b.data_ptr() + 4 * c0) if b is not None else None,
                                        y.data_ptr(), B, H, W, Cc, k, _stream()), "dwconv_fwd")
        ctx.save_for_backward(x)
        ctx.params = (w, b, c0)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        w, b, c0 = ctx.params
        B, H, W, Cc = x.shape
        k = w.shape[-1]
        dy = dy.contiguous()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        pg = _ParamGrads(w, b)
        gw = pg[0]
        gb = pg[1] if b is not None else torch.zeros(w.shape[0], device=x.device)
        L.check(L.lib().sast_dwconv_bwd(x.data_ptr(), w.data_ptr() + 4 * c0 * k * k, dy.data_ptr(), _ptr(dx), gw.data_ptr() + 4 * c0 * k * k,
                                        gb.data_ptr() + 4 * c0, B, H, W, Cc, k, _stream()), "dwconv_bwd")
        return (dx,) + pg.out() + (None,)


def dwconv(x_nhwc: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], c0: int = 0) -> torch.Tensor:
    """x (B,H,W,C) fp32 NHWC, w [Cw,1,k,k] (channels c0 .. c0+C-1 used), b [Cw] or None -> (B,H,W,C); parameter gradients accumulate
    in place into w.grad / b.grad (module contract, see the header)"""
    return _DwConv.apply(x_nhwc, w, b, int(c0))


# ---------------------------------------------------------------------------------------------- a13
class BnHandle:
    """what the conv that consumes a conv_bn_silu output needs to fold that producer's BatchNorm-backward reduction into its
    own dX epilogue (include/sast_hip.h: SastConvBnArgs.p_*), plus the flag telling the producer's backward it was done."""
    __slots__ = ("conv_out", "stats", "bn_w", "bn_b", "bn_ws", "cout", "red_done")

    def __init__(self):
        self.conv_out = self.stats = self.bn_w = self.bn_b = self.bn_ws = None
        self.cout = -1
        self.red_done = False

    def fill(self, conv_out, stats, bn_w, bn_b, bn_ws, cout):
        self.conv_out, self.stats, self.bn_w, self.bn_b, self.bn_ws, self.cout = conv_out, stats, bn_w, bn_b, bn_ws, cout
        return self


class SyncBatchNormGroup:
    """SyncBatchNorm for the conv + BatchNorm + SiLU units: the reference trains with `sync_batchnorm=True` whenever it runs DDP
    (train.py:167 -> torch.nn.SyncBatchNorm: batch statistics over the rows of ALL ranks, affine gradients local).  One object is shared
    by the BaseConvs of a model (`sast_amd.detection.convert_sync_batchnorm`, or found from the `torch.nn.SyncBatchNorm` modules that
    `torch.nn.SyncBatchNorm.convert_sync_batchnorm` / Lightning's `Trainer(sync_batchnorm=True)` left behind: `sync_group_for`); the C
    entry points are called in two phases around the all-reduces issued here (include/sast_hip.h: SastConvBnArgs.sync_phase).  On the
    RCCL backend ("nccl") the statistics all-reduces are captured INTO the step's hipGraph like any kernel node (`capturable()`,
    training.TrainStep.capture); a host-side backend (gloo: plumbing tests) cannot be captured and runs its PAFPN / head eagerly.
    `force=True` keeps the two-phase path on with ONE rank (the all-reduce is then the identity): the captured path can be exercised on
    a single GPU.

    Communicator: on RCCL the statistics all-reduces run on a PRIVATE communicator over the same ranks (`dist.new_group`, created at the
    first collective, which every rank reaches at the same point of the same model code).  The gradient buckets of the segmented step
    are all-reduced eagerly on a side stream WHILE the captured statistics all-reduces of the next segment replay on the main stream;
    two collectives in flight on ONE communicator from two streams have no defined order across ranks (hang or mixed-up sums), on two
    communicators they are independent."""

    def __init__(self, process_group=None, force: bool = False):
        import torch.distributed as dist
        self.dist, self.group = dist, process_group
        self._world = None          # resolved on first use: convert_sync_batchnorm may run before init_process_group
        self._ratio = None
        self._comm = None           # the communicator the statistics travel on (see the class docstring); False = use self.group
        self.force = bool(force)
        self.n_collectives = 0
        self._epoch = None          # identity of the default process group the cached world / communicator / ratio belong to

    def _validate(self):
        """destroy_process_group() + a new init_process_group() in the same process (Lightning fit -> test, the test suite) leaves this
        object -- cached in `_SYNC_GROUPS`, installed on the modules -- with the world size, the private communicator and the sample
        ratio of a group that no longer exists: drop them whenever the default group is gone or is another object (round-5 advice)"""
        d = self.dist
        cur = d.group.WORLD if d.is_initialized() else None
        if cur is not self._epoch:
            self._world = self._ratio = self._comm = None
            self._epoch = cur

    @property
    def world(self) -> int:
        self._validate()
        if self._world is None:
            if not self.dist.is_initialized():
                return 1            # not cached: a later init_process_group must still switch the group on
            self._world = self.dist.get_world_size(self.group)
        return self._world

    def active(self) -> bool:
        return self.world > 1 or self.force

    def capturable(self) -> bool:
        """True when every collective of a pass can sit inside a stream capture: RCCL enqueues its kernels on a HIP stream (torch's
        ProcessGroupNCCL is capture-aware); without a process group (forced, one rank) there is nothing to enqueue"""
        if not self.dist.is_initialized():
            return True
        return self.dist.get_backend(self.group) == "nccl"

    def communicator(self):
        """the process group the statistics all-reduces are issued on (created once, at the first eager collective).
        SAST_SYNC_BN_PRIVATE_COMM=0: the gradient buckets' communicator (then the caller must order the side-stream bucket all-reduce
        and the next segment's statistics all-reduces itself -- UNVERIFIED beyond one rank: no multi-GPU node was available to any round)"""
        self._validate()
        if self._comm is None:
            d = self.dist
            if d.get_backend(self.group) == "nccl" and os.environ.get("SAST_SYNC_BN_PRIVATE_COMM", "1") != "0":
                if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("sast_amd: SyncBatchNormGroup needs one eager pass before a hipGraph capture (it creates the "
                                       "private RCCL communicator of the statistics all-reduces)")
                ranks = d.get_process_group_ranks(self.group if self.group is not None else d.group.WORLD)
                self._comm = d.new_group(ranks=ranks, backend="nccl", use_local_synchronization=True)
            else:
                self._comm = False
        return self.group if self._comm is False else self._comm

    def exchange_batch(self, n_local: int, device):
        """once per pass over a model: every BatchNorm of the pass sees rows = samples * H_out * W_out, so the rows of all ranks follow
        from the SAMPLE counts (they differ between ranks when a step keeps only the labelled samples, modules/detection.py:161-171).
        Inside a stream capture the counts are those of the eager pass that preceded it: a replayed graph has static shapes on every
        rank, and reading the sum back is a host synchronisation a capture does not allow.
        -> the pass token (group, local samples): the PAFPN tags its outputs with it, and a head that is handed those very tensors takes
        the exchange over (`same_pass`) instead of repeating it."""
        if device.type == "cuda" and torch.cuda.is_current_stream_capturing():
            if self._ratio is None or self._ratio[1] != int(n_local):
                raise RuntimeError("sast_amd: SyncBatchNormGroup inside a hipGraph capture needs one eager pass with the same per-rank "
                                   "batch first (it fixes the sample counts of all ranks)")
            return (self, int(n_local))
        if self.world > 1:
            t = torch.tensor([float(n_local)], dtype=torch.float64, device=device)
            self.dist.all_reduce(t, group=self.communicator())
            self.n_collectives += 1
            total = int(round(float(t.item())))
        else:
            total = int(n_local)
        self._ratio = (total, int(n_local))
        return (self, int(n_local))

    def same_pass(self, x) -> bool:
        """the YOLOX head runs right behind the PAFPN: when its input IS a tensor the PAFPN tagged in this pass (`_sast_sync_pass`), the
        sample counts of all ranks are the ones already exchanged -- one host sync less per step.  The decision depends on the model code
        only (is the head fed the PAFPN's outputs directly?), never on a rank's data, so every rank takes the same branch; a head fed
        anything else (re-batched, filtered, or through a layout wrapper that made new tensor objects) does its own exchange."""
        tok = getattr(x, "_sast_sync_pass", None)
        return tok is not None and tok[0] is self and self._ratio is not None and tok[1] == self._ratio[1] == int(x.shape[0])

    def rows_total(self, m_local: int, batch_local: int) -> int:
        if self._ratio is None or self._ratio[1] != batch_local:
            raise RuntimeError("sast_amd: SyncBatchNormGroup.exchange_batch(batch) must be called at the start of the pass")
        return m_local // batch_local * self._ratio[0]

    def all_reduce(self, t):
        if self.dist.is_initialized():
            self.dist.all_reduce(t, group=self.communicator())
        self.n_collectives += 1

    def all_reduce_many(self, ts):
        """the statistics of several independent units in ONE collective (coalesced: one launch / one wire round for all tensors)"""
        if len(ts) == 1:
            return self.all_reduce(ts[0])
        if self.dist.is_initialized():
            comm = self.communicator()
            if self.dist.get_backend(self.group) == "nccl":      # ncclGroupStart / End around the per-tensor calls: one launch
                import warnings
                coalesced = getattr(self.dist, "all_reduce_coalesced", None)      # deprecated in torch: may disappear
                if coalesced is not None:
                    try:
                        with warnings.catch_warnings():
                            warnings.simplefilter("ignore")      # (torch marks the entry point deprecated; ProcessGroupNCCL implements it)
                            coalesced(list(ts), group=comm)
                        self.n_collectives += 1
                        return
                    except (AttributeError, NotImplementedError, TypeError):
                        pass                                     # fall through to the flat buffer (capture-safe: cat, all_reduce, copies)
            # host-side backends (gloo: the plumbing tests) and the fallback: one flat buffer (same dtype per call: the fp64 forward sums
            # or the fp32 backward sums)
            flat = torch.cat([t.reshape(-1) for t in ts])
            self.dist.all_reduce(flat, group=comm)
            torch._foreach_copy_([t.reshape(-1) for t in ts], list(flat.split([t.numel() for t in ts])))
        self.n_collectives += 1


_SYNC_GROUPS = {}


def sync_group_for(process_group=None) -> SyncBatchNormGroup:
    """the SyncBatchNormGroup standing for `torch.nn.SyncBatchNorm(process_group=...)` modules: one per process group, shared by
    every unit whose BatchNorm was converted by torch (`torch.nn.SyncBatchNorm.convert_sync_batchnorm`, train.py:167)"""
    key = id(process_group) if process_group is not None else None
    g = _SYNC_GROUPS.get(key)
    if g is None:
        g = _SYNC_GROUPS[key] = SyncBatchNormGroup(process_group)
    return g


def _bn_ws_blocks(bn_ws, Cout):
    """(fp64 forward sums, fp32 backward sums) views of one conv's reduction scratch (include/sast_hip.h: SAST_BN_WS_FLOATS)"""
    n = bn_ws.numel() // 6
    raw = bn_ws.data         # the scratch is saved for the backward (and is a slice of an arena other units saved too): the collectives
    return raw[:4 * n].view(torch.float64), raw[4 * n:]      # write it like the kernels do, outside autograd's version counting


def _cbs_fwd_args(x, x2, w, bn_w, bn_b, run_mean, run_var, ksize, stride, training, momentum, eps, bn_ws):
    """the argument block of one conv + BatchNorm + SiLU forward and what its backward keeps: (SastConvBnArgs, state dict)"""
    _need_gpu(x, w)
    x = x.contiguous()
    if not is_channels_last_weight(w):
        raise RuntimeError("sast_amd: conv weights must be stored channels_last ([Cout][KH][KW][Cin])")
    B, H, W, Cin = x.shape
    Cin1 = Cin
    if x2 is not None:       # virtual channel concat [x | x2] (1x1 convs): read in place, never materialised
        if ksize != 1 or stride != 1 or x2.shape[:3] != x.shape[:3]:
            raise RuntimeError("sast_amd: a two-source input is supported for 1x1 stride-1 convs of equal spatial size")
        x2 = x2.contiguous()
        Cin = Cin1 + x2.shape[-1]
    Cout = w.shape[0]
    groups = _conv_groups(w, Cin, x2)
    pad = (ksize - 1) // 2
    Ho, Wo = (H + 2 * pad - ksize) // stride + 1, (W + 2 * pad - ksize) // stride + 1
    M = B * Ho * Wo
    dev = x.device
    conv_out = torch.empty(M, Cout, device=dev)
    stats = torch.empty(2 * Cout, device=dev)
    y = torch.empty(B, Ho, Wo, Cout, device=dev)
    if bn_ws is None:  # zero-filled reduction scratch (a whole FPN passes slices of one arena: one memset per step)
        bn_ws = torch.zeros(bn_ws_floats(Cout), device=dev)
    a = _fill(L.SastConvBnArgs(), B=B, H=H, W=W, Cin=Cin, Cout=Cout, ksize=ksize, stride=stride, training=int(training), ldx=Cin1,
              ldy=Cout, bn_ws_zeroed=1, momentum=momentum, eps=eps, x=x, w=w, bn_w=bn_w, bn_b=bn_b, run_mean=_ptr(run_mean),
              run_var=_ptr(run_var), conv_out=conv_out, stats=stats, y=y, bn_ws=bn_ws, x2=_ptr(x2), Cin1=Cin1, ldx2=Cin - Cin1, groups=groups)
    st = dict(saved=(x, x2, conv_out, stats, bn_ws), params=(w, bn_w, bn_b), groups=groups, y=y,
              meta=(B, H, W, Cin, Cin1, Cout, ksize, stride, int(training), momentum, eps, M))
    return a, st


def _cbs_fwd_links(st, training, producers, handle):
    """the BatchNorm-backward folding links of one unit: (own handle filled, producers this unit's dX epilogue can serve)"""
    x, x2, conv_out, stats, bn_ws = st["saved"]
    w, bn_w, bn_b = st["params"]
    B, H, W, Cin, Cin1, Cout, ksize, stride = st["meta"][:8]
    own = handle.fill(conv_out, stats, bn_w, bn_b, bn_ws, Cout) if (training and handle is not None) else None
    p1, p2 = producers if (training and stride == 1 and st["groups"] == 1) else (None, None)   # (the depth-wise stencil has no dX epilogue to fold into)
    if p1 is not None and p1.cout != Cin1:
        p1 = None
    if p2 is not None and (x2 is None or p2.cout != Cin - Cin1):
        p2 = None
    return own, (p1, p2)


def _cbs_bwd_args(saved, params, meta, groups, own, producers, need1, need2, dy, dy2):
    """the argument block of one unit's backward: (SastConvBnArgs, dx, dx2, _ParamGrads, producers served, tensors to keep alive)"""
    x, x2, conv_out, stats, bn_ws = saved
    if dy is None:
        dy, dy2 = dy2, None
    if dy is None:
        dy = torch.zeros(conv_out.shape, device=conv_out.device)
    dy2 = dy2.contiguous() if dy2 is not None else None
    w, bn_w, bn_b = params
    B, H, W, Cin, Cin1, Cout, ksize, stride, training, momentum, eps, M = meta
    dy = dy.contiguous()
    need = need1 or (x2 is not None and need2)
    dx = torch.empty_like(x) if need else None
    dx2 = torch.empty_like(x2) if (need and x2 is not None) else None
    ws = torch.empty(M * Cout, device=x.device)
    p1, p2 = producers if need else (None, None)
    if p1 is not None and p1.red_done:
        p1 = None
    if p2 is not None and (p2.red_done or dx2 is None):
        p2 = None
    pk = {}
    for pre, h in (("p_", p1), ("p2_", p2)):
        if h is not None:
            pk.update({pre + "conv_out": h.conv_out, pre + "stats": h.stats, pre + "bn_w": h.bn_w, pre + "bn_b": h.bn_b,
                       pre + "bn_ws": h.bn_ws})
    pg = _ParamGrads(w, bn_w, bn_b)
    a = _fill(L.SastConvBnArgs(), B=B, H=H, W=W, Cin=Cin, Cout=Cout, ksize=ksize, stride=stride, training=training, ldx=Cin1,
              ldy=Cout, lddy=Cout, lddx=Cin1, bn_ws_zeroed=1, bn_red_done=int(own is not None and own.red_done), momentum=momentum,
              eps=eps, x=x, w=w, bn_w=bn_w, bn_b=bn_b, conv_out=conv_out, stats=stats, dy=dy, dx=_ptr(dx), dw=pg[0],
              d_bn_w=pg[1], d_bn_b=pg[2], bn_ws=bn_ws, ws=ws, x2=_ptr(x2), dx2=_ptr(dx2), Cin1=Cin1, ldx2=Cin - Cin1, dy2=_ptr(dy2),
              groups=groups, **pk)
    return a, dx, dx2, pg, (p1, p2), (dy, dy2, ws)


class _ConvBnSilu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, x2, w, bn_w, bn_b, run_mean, run_var, ksize, stride, training, momentum, eps, bn_ws, producers, handle, two_y,
                sync=None):
        a, st = _cbs_fwd_args(x, x2, w, bn_w, bn_b, run_mean, run_var, ksize, stride, training, momentum, eps, bn_ws)
        M, B, Cout = st["meta"][11], st["meta"][0], st["meta"][5]
        bn_ws = st["saved"][4]
        m_total = 0
        if sync is not None and training and sync.active():
            a.sync_phase = 1
            L.check(L.lib().sast_conv_bn_silu_fwd(C.byref(a), _stream()), "conv_bn_silu_fwd")
            sync.all_reduce(_bn_ws_blocks(bn_ws, Cout)[0])
            m_total = sync.rows_total(M, B)
            a.sync_phase, a.m_total = 2, m_total
        L.check(L.lib().sast_conv_bn_silu_fwd(C.byref(a), _stream()), "conv_bn_silu_fwd")
        ctx.sync = (sync, m_total) if m_total else None
        ctx.save_for_backward(*st["saved"])
        ctx.params, ctx.meta, ctx.groups = st["params"], st["meta"], st["groups"]
        ctx.handle, ctx.producers = _cbs_fwd_links(st, training, producers, handle)
        ctx.two_y = bool(two_y)
        y = st["y"]
        return (y, y.view_as(y)) if two_y else y     # two aliases for two consumers: see _LSTM.forward

    @staticmethod
    def backward(ctx, dy, dy2=None):
        _consume(ctx, "conv_bn_silu")
        Cout = ctx.meta[5]
        a, dx, dx2, pg, served, _keep = _cbs_bwd_args(ctx.saved_tensors, ctx.params, ctx.meta, ctx.groups, ctx.handle, ctx.producers,
                                                      ctx.needs_input_grad[0], ctx.needs_input_grad[1], dy, dy2)
        if ctx.sync is not None:
            sync, m_total = ctx.sync
            a.sync_phase = 1
            L.check(L.lib().sast_conv_bn_silu_bwd(C.byref(a), _stream()), "conv_bn_silu_bwd")
            # (phase 1 also added this process's (sum dz, sum dz * xhat) to d_bn_b / d_bn_w: the affine gradients stay local)
            sync.all_reduce(_bn_ws_blocks(ctx.saved_tensors[4], Cout)[1])
            a.sync_phase, a.m_total, a.d_bn_w, a.d_bn_b = 2, m_total, None, None
        L.check(L.lib().sast_conv_bn_silu_bwd(C.byref(a), _stream()), "conv_bn_silu_bwd")
        _dw_hold((ctx.saved_tensors, _keep, pg))
        for h in served:
            if h is not None:
                h.red_done = True
        return (dx, dx2) + pg.out() + (None,) * 12


_CBS_UNIT_IN = 7      # tensors per unit of _ConvBnSiluSyncGroup.apply: x, x2, w, bn_w, bn_b, running_mean, running_var


class _ConvBnSiluSyncGroup(torch.autograd.Function):
    """SEVERAL independent conv + BatchNorm + SiLU units under SyncBatchNorm as one autograd node: phase 1 of every unit, ONE
    (coalesced) statistics all-reduce, phase 2 of every unit -- forward and backward.  The dependent chain of a multi-rank step is the
    number of collectives, not of units: CSPLayer.conv1 / conv2 (same input), the three levels of the YOLOX head (stems; first and second
    3x3 of both towers) are such sets.  `units`: one tuple (ksize, stride, momentum, eps, bn_ws, producers, handle, two_y) per unit;
    `tensors`: _CBS_UNIT_IN per unit.  Training mode only (the caller checked `sync.active()`)."""

    @staticmethod
    def forward(ctx, sync, units, *tensors):
        n = len(units)
        jobs, saved, outs = [], [], []
        for i, (ksize, stride, momentum, eps, bn_ws, producers, handle, two_y) in enumerate(units):
            x, x2, w, bn_w, bn_b, rm, rv = tensors[_CBS_UNIT_IN * i:_CBS_UNIT_IN * (i + 1)]
            a, st = _cbs_fwd_args(x, x2, w, bn_w, bn_b, rm, rv, ksize, stride, True, momentum, eps, bn_ws)
            a.sync_phase = 1
            L.check(L.lib().sast_conv_bn_silu_fwd(C.byref(a), _stream()), "conv_bn_silu_fwd")
            jobs.append((a, st))
        sync.all_reduce_many([_bn_ws_blocks(st["saved"][4], st["meta"][5])[0] for _a, st in jobs])
        ctx.units = []
        for (a, st), (ksize, stride, momentum, eps, bn_ws, producers, handle, two_y) in zip(jobs, units):
            m_total = sync.rows_total(st["meta"][11], st["meta"][0])
            a.sync_phase, a.m_total = 2, m_total
            L.check(L.lib().sast_conv_bn_silu_fwd(C.byref(a), _stream()), "conv_bn_silu_fwd")
            own, prods = _cbs_fwd_links(st, True, producers, handle)
            ctx.units.append((st["params"], st["meta"], st["groups"], own, prods, bool(two_y), m_total, [t is not None for t in st["saved"]]))
            saved.extend(t for t in st["saved"] if t is not None)
            y = st["y"]
            outs.extend((y, y.view_as(y)) if two_y else (y,))
        ctx.sync = sync
        ctx.save_for_backward(*saved)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dys):
        _consume(ctx, "conv_bn_silu (SyncBatchNorm group)")
        sv, k, d = list(ctx.saved_tensors), 0, 0
        jobs = []
        for i, (params, meta, groups, own, prods, two_y, m_total, present) in enumerate(ctx.units):
            saved = []
            for has in present:
                saved.append(sv[k] if has else None)
                k += has
            dy, dy2 = (dys[d], dys[d + 1]) if two_y else (dys[d], None)
            d += 2 if two_y else 1
            a, dx, dx2, pg, served, keep = _cbs_bwd_args(tuple(saved), params, meta, groups, own, prods,
                                                         ctx.needs_input_grad[2 + _CBS_UNIT_IN * i], ctx.needs_input_grad[3 + _CBS_UNIT_IN * i], dy, dy2)
            a.sync_phase = 1
            L.check(L.lib().sast_conv_bn_silu_bwd(C.byref(a), _stream()), "conv_bn_silu_bwd")
            jobs.append((a, dx, dx2, pg, served, keep, saved[4], meta[5], m_total))
        ctx.sync.all_reduce_many([_bn_ws_blocks(ws, cout)[1] for (_a, _dx, _dx2, _pg, _s, _k, ws, cout, _m) in jobs])
        grads = [None, None]
        for a, dx, dx2, pg, served, _keep, _ws, _cout, m_total in jobs:
            a.sync_phase, a.m_total, a.d_bn_w, a.d_bn_b = 2, m_total, None, None
            L.check(L.lib().sast_conv_bn_silu_bwd(C.byref(a), _stream()), "conv_bn_silu_bwd")
            for h in served:
                if h is not None:
                    h.red_done = True
            grads.extend((dx, dx2) + pg.out() + (None, None))
        _dw_hold((sv, jobs))
        return tuple(grads)


def conv_bn_silu_sync_group(sync: "SyncBatchNormGroup", items):
    """training-mode conv + BatchNorm + SiLU of SEVERAL independent units whose batch statistics span the ranks of `sync`, with one
    statistics all-reduce per direction for all of them.  items: one dict per unit with the arguments of `conv_bn_silu` (x_nhwc, w, bn_w,
    bn_b, run_mean, run_var, ksize, stride, momentum, eps, bn_ws, sole_consumer, two_outputs).  -> one output per unit (a (y, alias)
    pair where two_outputs)."""
    units, tensors = [], []
    for it in items:
        xin = it["x_nhwc"]
        x, x2 = xin if isinstance(xin, (tuple, list)) else (xin, None)
        two = bool(it.get("two_outputs", False)) and TWO_OUT and torch.is_grad_enabled()
        handle = None if two else BnHandle()
        units.append((int(it["ksize"]), int(it["stride"]), float(it.get("momentum", 0.1)), float(it.get("eps", 1e-5)), it.get("bn_ws"),
                      _producers(x, x2, it.get("sole_consumer", False)), handle, two))
        tensors.extend((x, x2, it["w"], it["bn_w"], it["bn_b"], it["run_mean"], it["run_var"]))
    flat = _ConvBnSiluSyncGroup.apply(sync, units, *tensors)
    outs, k = [], 0
    for it, u in zip(items, units):
        if u[7]:
            outs.append((flat[k], flat[k + 1]))
            k += 2
        else:
            y = flat[k]
            y._sast_bn = u[6]      # lets a sole consumer of y fold this conv's BatchNorm-backward reduction into its dX epilogue
            k += 1
            outs.append((y, y) if it.get("two_outputs", False) else y)
    return outs


def _conv_groups(w, cin: int, x2=None) -> int:
    """1 for a dense conv weight (Cout, Cin, k, k); Cin for the depth-wise weight (C, 1, k, k) of YOLOX's DWConv.dconv
    (network_blocks.py:57-76).  Other group counts do not occur in the reference."""
    if w.shape[1] == cin:
        return 1
    if w.shape[1] == 1 and w.shape[0] == cin and x2 is None:
        return cin
    raise RuntimeError(f"sast_amd: conv weight {tuple(w.shape)} fits neither a dense conv of {cin} input channels nor a depth-wise one")


def bn_ws_floats(cout: int) -> int:
    """floats of BatchNorm reduction scratch one conv_bn_silu call consumes (include/sast_hip.h: SAST_BN_WS_FLOATS)"""
    return int(L.lib().sast_conv_bn_ws_floats(int(cout)))


class _ConvBnSilu2(torch.autograd.Function):
    """two 1x1 stride-1 conv + BatchNorm(batch statistics) + SiLU of the SAME input (CSPLayer.conv1 / conv2) as one GEMM over
    the stacked weights; the backward's dX is the sum of both input gradients (include/sast_hip.h: SastConvBn2Args)."""

    @staticmethod
    def forward(ctx, x, x2, w0, bnw0, bnb0, rm0, rv0, w1, bnw1, bnb1, rm1, rv1, mom0, eps0, mom1, eps1, ws0, ws1, producers, handles, ksize):
        _need_gpu(x, w0, w1)
        x = x.contiguous()
        for w in (w0, w1):
            if not is_channels_last_weight(w):
                raise RuntimeError("sast_amd: conv weights must be stored channels_last ([Cout][KH][KW][Cin])")
        B, H, W, Cin1 = x.shape
        Cin = Cin1
        if x2 is not None:
            if x2.shape[:3] != x.shape[:3]:
                raise RuntimeError("sast_amd: the two sources of a virtual concat need equal spatial size")
            x2 = x2.contiguous()
            Cin = Cin1 + x2.shape[-1]
        Cout = w0.shape[0]
        if w1.shape[0] != Cout or tuple(w0.shape[1:]) != (Cin, ksize, ksize) or tuple(w1.shape[1:]) != (Cin, ksize, ksize) or \
                ksize not in (1, 3) or (ksize == 3 and x2 is not None):
            raise RuntimeError("sast_amd: conv_bn_silu2 needs two 1x1 (or 3x3, single-source) convs of the same input with equal Cout")
        M, dev = B * H * W, x.device
        co = [torch.empty(M, Cout, device=dev) for _ in range(2)]
        st = [torch.empty(2 * Cout, device=dev) for _ in range(2)]
        ys = [torch.empty(B, H, W, Cout, device=dev) for _ in range(2)]
        if ws0 is None or ws1 is None:
            ws0, ws1 = torch.zeros(bn_ws_floats(Cout), device=dev), torch.zeros(bn_ws_floats(Cout), device=dev)
        a = _fill(L.SastConvBn2Args(), B=B, H=H, W=W, Cin=Cin, Cout=Cout, ldx=Cin1, Cin1=Cin1, ldx2=Cin - Cin1, bn_ws_zeroed=1, training=1, ksize=ksize,
                  momentum0=mom0, momentum1=mom1, eps0=eps0, eps1=eps1, x=x, x2=_ptr(x2), w0=w0, w1=w1, bn_w0=bnw0, bn_w1=bnw1,
                  bn_b0=bnb0, bn_b1=bnb1, run_mean0=_ptr(rm0), run_mean1=_ptr(rm1), run_var0=_ptr(rv0), run_var1=_ptr(rv1),
                  conv_out0=co[0], conv_out1=co[1], stats0=st[0], stats1=st[1], y0=ys[0], y1=ys[1], bn_ws0=ws0, bn_ws1=ws1)
        L.check(L.lib().sast_conv_bn_silu2_fwd(C.byref(a), _stream()), "conv_bn_silu2_fwd")
        ctx.save_for_backward(x, x2, co[0], co[1], st[0], st[1], ws0, ws1)
        ctx.params = (w0, bnw0, bnb0, w1, bnw1, bnb1)
        ctx.meta = (B, H, W, Cin, Cin1, Cout, mom0, eps0, mom1, eps1, M, ksize)
        ctx.handles = (handles[0].fill(co[0], st[0], bnw0, bnb0, ws0, Cout), handles[1].fill(co[1], st[1], bnw1, bnb1, ws1, Cout))
        p1, p2 = producers
        if p1 is not None and p1.cout != Cin1:
            p1 = None
        if p2 is not None and (x2 is None or p2.cout != Cin - Cin1):
            p2 = None
        ctx.producers = (p1, p2)
        return ys[0], ys[1]

    @staticmethod
    def backward(ctx, dy0, dy1):
        x, x2, co0, co1, st0, st1, ws0, ws1 = ctx.saved_tensors
        _consume(ctx, "conv_bn_silu2")
        w0, bnw0, bnb0, w1, bnw1, bnb1 = ctx.params
        B, H, W, Cin, Cin1, Cout, mom0, eps0, mom1, eps1, M, ksize = ctx.meta
        dy0 = dy0.contiguous() if dy0 is not None else torch.zeros(B, H, W, Cout, device=x.device)
        dy1 = dy1.contiguous() if dy1 is not None else torch.zeros(B, H, W, Cout, device=x.device)
        need = ctx.needs_input_grad[0] or (x2 is not None and ctx.needs_input_grad[1])
        dx = torch.empty_like(x) if need else None
        dx2 = torch.empty_like(x2) if (need and x2 is not None) else None
        dws = torch.empty(M, 2 * Cout, device=x.device)      # rows [dconv0 | dconv1]
        p1, p2 = ctx.producers if need else (None, None)
        if p1 is not None and p1.red_done:
            p1 = None
        if p2 is not None and (p2.red_done or dx2 is None):
            p2 = None
        pk = {}
        for pre, h in (("p_", p1), ("p2_", p2)):
            if h is not None:
                pk.update({pre + "conv_out": h.conv_out, pre + "stats": h.stats, pre + "bn_w": h.bn_w, pre + "bn_b": h.bn_b,
                           pre + "bn_ws": h.bn_ws})
        h0, h1 = ctx.handles
        pg = _ParamGrads(w0, bnw0, bnb0, w1, bnw1, bnb1)
        a = _fill(L.SastConvBn2Args(), B=B, H=H, W=W, Cin=Cin, Cout=Cout, ldx=Cin1, Cin1=Cin1, ldx2=Cin - Cin1, bn_ws_zeroed=1, training=1, ksize=ksize,
                  bn_red_done0=int(h0.red_done), bn_red_done1=int(h1.red_done), momentum0=mom0, momentum1=mom1, eps0=eps0, eps1=eps1,
                  x=x, x2=_ptr(x2), w0=w0, w1=w1, bn_w0=bnw0, bn_w1=bnw1, bn_b0=bnb0, bn_b1=bnb1, conv_out0=co0, conv_out1=co1,
                  stats0=st0, stats1=st1, bn_ws0=ws0, bn_ws1=ws1, dy0=dy0, dy1=dy1, dw0=pg[0], dw1=pg[3], d_bn_w0=pg[1],
                  d_bn_w1=pg[4], d_bn_b0=pg[2], d_bn_b1=pg[5], ws0=dws, dx=_ptr(dx), dx2=_ptr(dx2), **pk)
        L.check(L.lib().sast_conv_bn_silu2_bwd(C.byref(a), _stream()), "conv_bn_silu2_bwd")
        _dw_hold((x, x2, co0, co1, st0, st1, ws0, ws1, dy0, dy1, dws, pg))
        for h in (p1, p2):
            if h is not None:
                h.red_done = True
        g = pg.out()        # forward(ctx, x, x2, w0, bnw0, bnb0, rm0, rv0, w1, bnw1, bnb1, rm1, rv1, ... 9 more)
        return (dx, dx2) + g[0:3] + (None, None) + g[3:6] + (None,) * 11


@torch.no_grad()
def conv_bn_silu2_infer(x_nhwc, conv0, conv1, ksize=1):
    """eval mode, no autograd: two convs (1x1, or 3x3 stride 1) of the same input + BatchNorm(running statistics) + SiLU in ONE
    launch over the stacked weights.  conv0 / conv1 as in conv_bn_silu2."""
    x, x2 = x_nhwc if isinstance(x_nhwc, (tuple, list)) else (x_nhwc, None)
    _need_gpu(x)
    x = x.contiguous()
    (w0, g0, b0, rm0, rv0, _m0, e0), (w1, g1, b1, rm1, rv1, _m1, e1) = conv0, conv1
    for w in (w0, w1):
        if not is_channels_last_weight(w):
            raise RuntimeError("sast_amd: conv weights must be stored channels_last ([Cout][KH][KW][Cin])")
    B, H, W, Cin1 = x.shape
    Cin = Cin1
    if x2 is not None:
        if ksize != 1 or x2.shape[:3] != x.shape[:3]:
            raise RuntimeError("sast_amd: a two-source input is supported for 1x1 convs of equal spatial size")
        x2 = x2.contiguous()
        Cin = Cin1 + x2.shape[-1]
    Cout = w0.shape[0]
    if w1.shape[0] != Cout or tuple(w0.shape[1:]) != (Cin, ksize, ksize) or tuple(w1.shape[1:]) != (Cin, ksize, ksize):
        raise RuntimeError("sast_amd: conv_bn_silu2_infer needs two convs of the same input with equal shapes")
    y0, y1 = torch.empty(B, H, W, Cout, device=x.device), torch.empty(B, H, W, Cout, device=x.device)
    a = _fill(L.SastConvBn2Args(), B=B, H=H, W=W, Cin=Cin, Cout=Cout, ldx=Cin1, Cin1=Cin1, ldx2=Cin - Cin1, training=0, ksize=ksize,
              eps0=float(e0), eps1=float(e1), x=x, x2=_ptr(x2), w0=w0, w1=w1, bn_w0=g0, bn_w1=g1, bn_b0=b0, bn_b1=b1, run_mean0=rm0,
              run_mean1=rm1, run_var0=rv0, run_var1=rv1, y0=y0, y1=y1)
    L.check(L.lib().sast_conv_bn_silu2_fwd(C.byref(a), _stream()), "conv_bn_silu2_fwd")
    return y0, y1


def conv_bn_silu2(x_nhwc, conv0, conv1, bn_ws=(None, None), sole_consumer=False, ksize=1):
    """(y0, y1) = two training-mode 1x1 (or 3x3 stride-1) conv + BatchNorm + SiLU of the same input in shared launches.  conv0 / conv1:
    (weight, bn_weight, bn_bias, running_mean, running_var, momentum, eps); x_nhwc a tensor or a pair standing for a channel
    concat; sole_consumer: nothing else consumes the input (see conv_bn_silu)."""
    x, x2 = x_nhwc if isinstance(x_nhwc, (tuple, list)) else (x_nhwc, None)
    prods = _producers(x, x2, sole_consumer)
    (w0, g0, b0, rm0, rv0, m0, e0), (w1, g1, b1, rm1, rv1, m1, e1) = conv0, conv1
    hs = (BnHandle(), BnHandle())
    y0, y1 = _ConvBnSilu2.apply(x, x2, w0, g0, b0, rm0, rv0, w1, g1, b1, rm1, rv1, float(m0), float(e0), float(m1), float(e1),
                                bn_ws[0], bn_ws[1], prods, hs, int(ksize))
    y0._sast_bn, y1._sast_bn = hs
    return y0, y1


def _producers(x, x2, sole):
    """BnHandles of the inputs this op is the sole consumer of; `sole`: bool, or one bool per source of a pair input"""
    s1, s2 = sole if isinstance(sole, (tuple, list)) else (sole, sole)
    if not BN_FOLD:
        return (None, None)
    return (getattr(x, "_sast_bn", None) if s1 else None, getattr(x2, "_sast_bn", None) if (s2 and x2 is not None) else None)
TWO_OUT = os.environ.get("SAST_TWO_OUT", "1") != "0"       # outputs with two consumers as two aliases (gradients summed inside the backward kernels)
CONV_PAIR = os.environ.get("SAST_CONV_PAIR", "1") != "0"   # CSPLayer.conv1 / conv2 (same input) through conv_bn_silu2
BN_FOLD = os.environ.get("SAST_BN_FOLD", "1") != "0"   # fold producers' BatchNorm-backward reductions into consumers' dX epilogues
SYNC_BN_GROUPS = os.environ.get("SAST_SYNC_BN_GROUPS", "1") != "0"   # SyncBatchNorm: independent units share one statistics all-reduce (CSP conv1 / conv2, head levels)


def conv_bn_silu(x_nhwc, w, bn_w, bn_b, run_mean, run_var, ksize, stride, training, momentum=0.1, eps=1e-5, bn_ws=None,
                 sole_consumer=False, two_outputs=False, sync=None):
    """x_nhwc: a tensor, or a pair (xa, xb) standing for their channel concat (1x1 convs; the concat is never built).
    bn_ws: optional zero-filled fp32[bn_ws_floats(Cout)] scratch (consumed: do not reuse within a step).
    sync: a SyncBatchNormGroup -- training-mode batch statistics over the rows of all its ranks."""
    x, x2 = x_nhwc if isinstance(x_nhwc, (tuple, list)) else (x_nhwc, None)
    if sync is not None and not (training and sync.active()):
        sync = None
    if two_outputs and (not TWO_OUT or not training or not torch.is_grad_enabled()):    # nothing to gain without a training-mode backward
        y = conv_bn_silu(x_nhwc, w, bn_w, bn_b, run_mean, run_var, ksize, stride, training, momentum, eps, bn_ws, sole_consumer, sync=sync)
        return y, y
    if not training and not (torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (x, x2, w, bn_w, bn_b))):
        return _conv_bn_silu_infer(x, x2, w, bn_w, bn_b, run_mean, run_var, ksize, stride, float(eps))
    # sole_consumer: the caller guarantees that this conv is the ONLY consumer of its input tensor(s); where such an input
    # is itself a conv_bn_silu output (it carries a BnHandle), that producer's BatchNorm-backward reduction is folded into
    # this conv's dX epilogue (one launch fewer per conv in the backward pass)
    prods = _producers(x, x2, sole_consumer)
    if two_outputs:     # (y, y_alias) for an output with two consumers: their gradients meet inside the BatchNorm-backward kernels
        return _ConvBnSilu.apply(x, x2, w, bn_w, bn_b, run_mean, run_var, ksize, stride, True, float(momentum), float(eps), bn_ws, prods,
                                 None, True, sync)
    handle = BnHandle() if training else None
    y = _ConvBnSilu.apply(x, x2, w, bn_w, bn_b, run_mean, run_var, ksize, stride, bool(training), float(momentum), float(eps), bn_ws, prods,
                          handle, False, sync)
    if handle is not None:
        y._sast_bn = handle      # lets a sole consumer of y fold this conv's BatchNorm-backward reduction into its dX epilogue
    return y


def _conv_bn_silu_infer(x, x2, w, bn_w, bn_b, run_mean, run_var, ksize, stride, eps):
    """eval mode, no autograd: conv + BatchNorm(running statistics) + SiLU in ONE launch (BN + SiLU in the GEMM epilogue)."""
    _need_gpu(x, w)
    x = x.contiguous()
    if not is_channels_last_weight(w):
        raise RuntimeError("sast_amd: conv weights must be stored channels_last ([Cout][KH][KW][Cin])")
    B, H, W, Cin = x.shape
    Cin1 = Cin
    if x2 is not None:
        if ksize != 1 or stride != 1 or x2.shape[:3] != x.shape[:3]:
            raise RuntimeError("sast_amd: a two-source input is supported for 1x1 stride-1 convs of equal spatial size")
        x2 = x2.contiguous()
        Cin = Cin1 + x2.shape[-1]
    Cout = w.shape[0]
    pad = (ksize - 1) // 2
    Ho, Wo = (H + 2 * pad - ksize) // stride + 1, (W + 2 * pad - ksize) // stride + 1
    y = torch.empty(B, Ho, Wo, Cout, device=x.device)
    a = _fill(L.SastConvBnArgs(), B=B, H=H, W=W, Cin=Cin, Cout=Cout, ksize=ksize, stride=stride, training=0, ldx=Cin1, ldy=Cout,
              bn_ws_zeroed=1, momentum=0.0, eps=eps, x=x, w=w, bn_w=bn_w, bn_b=bn_b, run_mean=run_mean, run_var=run_var, y=y,
              x2=_ptr(x2), Cin1=Cin1, ldx2=Cin - Cin1, groups=_conv_groups(w, Cin, x2))
    L.check(L.lib().sast_conv_bn_silu_fwd(C.byref(a), _stream()), "conv_bn_silu_fwd")
    return y


class _UpsampleCat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        _need_gpu(a, b)
        a, b = a.contiguous(), b.contiguous()
        B, H, W, C1 = a.shape
        C2 = b.shape[-1]
        assert b.shape[:3] == (B, 2 * H, 2 * W)
        out = torch.empty(B, 2 * H, 2 * W, C1 + C2, device=a.device)
        L.check(L.lib().sast_upsample_cat_fwd(a.data_ptr(), b.data_ptr(), out.data_ptr(), B, H, W, C1, C2, _stream()), "upsample_cat_fwd")
        ctx.meta = (B, H, W, C1, C2)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, H, W, C1, C2 = ctx.meta
        dout = dout.contiguous()
        da = torch.empty(B, H, W, C1, device=dout.device)
        db = torch.empty(B, 2 * H, 2 * W, C2, device=dout.device)
        L.check(L.lib().sast_upsample_cat_bwd(dout.data_ptr(), da.data_ptr(), db.data_ptr(), B, H, W, C1, C2, _stream()), "upsample_cat_bwd")
        return da, db


def upsample_cat(a, b):
    return _UpsampleCat.apply(a, b)


class _Cat2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        _need_gpu(a, b)
        a, b = a.contiguous(), b.contiguous()
        C1, C2 = a.shape[-1], b.shape[-1]
        rows = a.numel() // C1
        out = torch.empty(*a.shape[:-1], C1 + C2, device=a.device)
        L.check(L.lib().sast_cat2_fwd(a.data_ptr(), b.data_ptr(), out.data_ptr(), rows, C1, C2, _stream()), "cat2_fwd")
        ctx.meta = (a.shape, b.shape, rows, C1, C2)
        return out

    @staticmethod
    def backward(ctx, dout):
        sa, sb, rows, C1, C2 = ctx.meta
        dout = dout.contiguous()
        da, db = torch.empty(sa, device=dout.device), torch.empty(sb, device=dout.device)
        L.check(L.lib().sast_cat2_bwd(dout.data_ptr(), da.data_ptr(), db.data_ptr(), rows, C1, C2, _stream()), "cat2_bwd")
        return da, db


def cat2(a, b):
    return _Cat2.apply(a, b)


# ---------------------------------------------------------------------------------------------- (f)2 label-sparse gather
class _GatherSamples(torch.autograd.Function):
    """out = cat_t(x_t[idx_t]) over the timesteps of a sequence (BackboneFeatureSelector, modules/utils/detection.py:24-47)."""

    @staticmethod
    def forward(ctx, table, *xs):
        _need_gpu(*xs)
        xs = tuple(x.contiguous() for x in xs)
        B = xs[0].shape[0]
        sample = xs[0][0].numel()
        if any(x.shape != xs[0].shape or x.dtype != torch.float32 for x in xs):
            raise RuntimeError("sast_amd: gather_samples needs fp32 tensors of one shape (one feature map over the timesteps)")
        if len(xs) > 32 or len(table) > 256 or B > 256 or sample % 4:
            raise RuntimeError("sast_amd: gather_samples supports <= 32 timesteps, <= 256 selected samples, batch <= 256")
        out = torch.empty((len(table),) + tuple(xs[0].shape[1:]), device=xs[0].device)
        a = L.SastSampleGather()
        a.n_src, a.n_out, a.B, a.sample_floats, a.out = len(xs), len(table), B, sample, out.data_ptr()
        for t, x in enumerate(xs):
            a.src[t] = x.data_ptr()
        for j, (t, b) in enumerate(table):
            a.t_of[j], a.b_of[j] = t, b
        L.check(L.lib().sast_gather_samples(C.byref(a), _stream()), "gather_samples")
        ctx.table, ctx.meta = table, (len(xs), B, sample, tuple(xs[0].shape))
        return out

    @staticmethod
    def backward(ctx, dout):
        n_src, B, sample, shape = ctx.meta
        dout = dout.contiguous()
        dxs = tuple(torch.empty(shape, device=dout.device) for _ in range(n_src))
        a = L.SastSampleGather()
        a.n_src, a.n_out, a.B, a.sample_floats, a.out = n_src, len(ctx.table), B, sample, dout.data_ptr()
        for t, d in enumerate(dxs):
            a.dsrc[t] = d.data_ptr()
        for j, (t, b) in enumerate(ctx.table):
            a.t_of[j], a.b_of[j] = t, b
        L.check(L.lib().sast_gather_samples_bwd(C.byref(a), _stream()), "gather_samples_bwd")
        return (None,) + dxs


def gather_samples(xs, indices) -> torch.Tensor:
    """xs: the feature map of T timesteps, each (B, ...) fp32; indices: per timestep the list of selected batch indices (unique
    within a timestep, may be empty) -> (sum_t len(indices[t]), ...) = torch.cat([x[idx] for x, idx in zip(xs, indices) if idx])."""
    table = tuple((t, int(b)) for t, idx in enumerate(indices) for b in idx)
    for t, idx in enumerate(indices):
        if len(set(int(b) for b in idx)) != len(idx):
            raise RuntimeError("sast_amd: gather_samples needs unique batch indices per timestep")
    return _GatherSamples.apply(table, *xs)


@torch.no_grad()
def zero_samples(x: torch.Tensor, indices_or_bool=None) -> torch.Tensor:
    """RNNStates.recursive_reset (modules/utils/detection.py:96-116): x[indices_or_bool] = 0 in place (all samples when None)."""
    _need_gpu(x)
    if x.dtype != torch.float32 or not (x.is_contiguous() or x.permute(0, 2, 3, 1).is_contiguous()):
        raise RuntimeError("sast_amd: zero_samples needs a dense fp32 tensor with the batch as its outermost dimension")
    B = x.shape[0]
    m = L.SastSampleMask()
    if indices_or_bool is None:
        sel = range(B)
    else:
        t = torch.as_tensor(indices_or_bool).cpu()
        sel = torch.nonzero(t).view(-1).tolist() if t.dtype == torch.bool else [int(v) % B for v in t.view(-1).tolist()]
    for b in sel:
        m.sel[b] = 1
    L.check(L.lib().sast_zero_samples(x.data_ptr(), B, x[0].numel(), C.byref(m), _stream()), "zero_samples")
    return x


@torch.no_grad()
def adamw_onecycle_step(p, g, m, v, lr_step, sched, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, grad_scale=1.0, clip_value=0.0):
    """fused AdamW with the OneCycleLR learning rate evaluated on the device from the step counter lr_step[1]
    (sched: dist.OneCycleLR; modules/detection.py:418-431)."""
    _need_gpu(p, g, m, v, lr_step)
    L.check(L.lib().sast_adamw_onecycle(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr_step.data_ptr(), float(beta1),
                                        float(beta2), eps, weight_decay, grad_scale, clip_value, sched.initial_lr, sched.max_lr, sched.min_lr,
                                        sched.end1, sched.end2, _stream()), "adamw_onecycle")


@torch.no_grad()
def adamw_step(p, g, m, v, lr_step, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, grad_scale=1.0, clip_value=0.0):
    """fused AdamW on flat fp32 buffers; lr_step = device tensor [lr, step]."""
    _need_gpu(p, g, m, v, lr_step)
    L.check(L.lib().sast_adamw(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr_step.data_ptr(), float(beta1), float(beta2),
                               eps, weight_decay, grad_scale, clip_value, _stream()), "adamw")


# ---------------------------------------------------------------------------------------------- YOLOX head
class _HeadPredLoss(torch.autograd.Function):
    """prediction convs of all levels + SimOTA assignment + losses (yolo_head.py:184-246,291-606) as one autograd node.
    inputs per level: reg_feat, cls_feat (B,H,W,hidden NHWC), w_reg, b_reg, w_obj, b_obj, w_cls, b_cls.
    -> losses (6,) = loss, 5*iou, conf, cls, l1, num_fg ratio (only [0] carries a gradient) and the inference-style predictions."""

    @staticmethod
    def forward(ctx, labels, levels, num_classes, decode, use_l1, *t):
        nlev = len(levels)
        assert len(t) == 8 * nlev
        rf0 = t[0]
        _need_gpu(rf0, labels)
        dev = rf0.device
        B = rf0.shape[0]
        hid = rf0.shape[-1]
        geom = L.SastHeadGeom()
        geom.n_levels = nlev
        for k, (h, w, s) in enumerate(levels):
            geom.H[k], geom.W[k], geom.stride[k] = int(h), int(w), float(s)
        A = sum(int(h) * int(w) for h, w, _ in levels)
        no = 5 + num_classes
        labels = labels.to(device=dev, dtype=torch.float32).contiguous()
        G = labels.shape[1]
        pred = torch.empty(B, A, no, device=dev)
        train = torch.empty(B, A, no, device=dev)
        feats = []
        off = 0
        for k, (h, w, s) in enumerate(levels):
            rf, cf, w_reg, b_reg, w_obj, b_obj, w_cls, b_cls = t[8 * k:8 * k + 8]
            rf, cf = rf.contiguous(), cf.contiguous()
            feats.append((rf, cf))
            L.check(L.lib().sast_head_pred_fwd(rf.data_ptr(), cf.data_ptr(), w_reg.data_ptr(), b_reg.data_ptr(), w_obj.data_ptr(), b_obj.data_ptr(),
                                               w_cls.data_ptr(), b_cls.data_ptr(), pred.data_ptr(), train.data_ptr(), B, int(h), int(w), hid,
                                               num_classes, float(s), off, A, int(decode), _stream()), "head_pred_fwd")
            off += int(h) * int(w)
        losses = torch.empty(6, device=dev)
        draw = torch.empty(B, A, no, device=dev)
        fg = torch.empty(B, A, device=dev, dtype=torch.int32)
        mg = torch.empty(B, A, device=dev, dtype=torch.int32)
        piou = torch.empty(B, A, device=dev)
        ws = torch.empty(L.lib().sast_yolox_loss_ws_bytes(B, A, G), device=dev, dtype=torch.uint8)
        L.check(L.lib().sast_yolox_loss(train.data_ptr(), labels.data_ptr(), C.byref(geom), B, G, num_classes, int(use_l1), losses.data_ptr(), draw.data_ptr(),
                                        fg.data_ptr(), mg.data_ptr(), piou.data_ptr(), ws.data_ptr(), _stream()), "yolox_loss")
        ctx.save_for_backward(draw, *[x for pair in feats for x in pair])
        ctx.params = [t[8 * k + 2:8 * k + 8] for k in range(nlev)]
        ctx.meta = (levels, num_classes, B, A, hid)
        ctx.mark_non_differentiable(pred, fg, mg, piou)
        return losses, pred, fg, mg, piou

    @staticmethod
    def backward(ctx, dlosses, _dpred, _dfg, _dmg, _dpiou):
        draw = ctx.saved_tensors[0]
        feats = ctx.saved_tensors[1:]
        levels, num_classes, B, A, hid = ctx.meta
        draw = (draw * dlosses[0]).contiguous()          # only losses[0] (the total) is a training signal
        grads = [None, None, None, None, None]
        off = 0
        for k, (h, w, _s) in enumerate(levels):
            rf, cf = feats[2 * k], feats[2 * k + 1]
            w_reg, b_reg, w_obj, b_obj, w_cls, b_cls = ctx.params[k]
            drf, dcf = torch.empty_like(rf), torch.empty_like(cf)
            gp = _ParamGrads(w_reg, b_reg, w_obj, b_obj, w_cls, b_cls)      # held until the launch is enqueued
            L.check(L.lib().sast_head_pred_bwd(draw.data_ptr(), rf.data_ptr(), cf.data_ptr(), w_reg.data_ptr(), w_obj.data_ptr(), w_cls.data_ptr(),
                                               drf.data_ptr(), dcf.data_ptr(), gp[0].data_ptr(), gp[1].data_ptr(), gp[2].data_ptr(),
                                               gp[3].data_ptr(), gp[4].data_ptr(), gp[5].data_ptr(), B, int(h), int(w), hid,
                                               num_classes, off, A, _stream()), "head_pred_bwd")
            off += int(h) * int(w)
            grads += [drf, dcf] + list(gp.out())
        return tuple(grads)


def head_pred_loss(labels, levels, num_classes, decode, per_level_tensors, use_l1=False):
    """levels: [(H, W, stride)]; per_level_tensors: [(reg_feat, cls_feat, w_reg, b_reg, w_obj, b_obj, w_cls, b_cls)]
    -> (losses(6,), pred, fg_mask (B,A) int32, matched_gt (B,A) int32 (-1 = background), matched_iou (B,A))"""
    flat = [x for lv in per_level_tensors for x in lv]
    return _HeadPredLoss.apply(labels, tuple((int(h), int(w), float(s)) for h, w, s in levels), int(num_classes), bool(decode), bool(use_l1), *flat)


@torch.no_grad()
def postprocess(prediction: torch.Tensor, num_classes: int, conf_thre: float = 0.7, nms_thre: float = 0.45, class_agnostic: bool = False):
    """yolox/utils/boxes.py:32-76: prediction (B, A, 5+nc) from the head -> list of (n_i, 7) tensors (x1, y1, x2, y2, obj_conf,
    class_conf, class_pred) by decreasing score, None for images without detections.  One host sync (the detection counts)."""
    _need_gpu(prediction)
    prediction = prediction.float().contiguous()
    B, A, no = prediction.shape
    assert no == 5 + num_classes
    out = torch.empty(B, A, 7, device=prediction.device)
    n_out = torch.empty(B, device=prediction.device, dtype=torch.int32)
    ws = torch.empty(L.lib().sast_postprocess_ws_bytes(B, A), device=prediction.device, dtype=torch.uint8)
    L.check(L.lib().sast_postprocess(prediction.data_ptr(), B, A, num_classes, float(conf_thre), float(nms_thre), int(bool(class_agnostic)), out.data_ptr(),
                                     n_out.data_ptr(), ws.data_ptr(), _stream()), "postprocess")
    counts = n_out.tolist()
    return [out[b, :n].clone() if n else None for b, n in enumerate(counts)]
