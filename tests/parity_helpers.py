"""Comparison helpers shared by the GPU parity tests and the CPU test of the kink-aware gradient comparison."""
import os

import torch

GRAD_RTOL = 3e-4


def _test_id():
    return os.environ.get("PYTEST_CURRENT_TEST", "?").split("::")[-1].split(" ")[0]


def maxnorm_close(a, b, rtol, what=""):
    from conftest import record_error
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    scale = float(b.abs().max()) + 1e-12
    err = float((a - b).abs().max())
    record_error(_test_id(), what, err, scale, rtol)
    assert err <= rtol * scale + 1e-9, f"{what}: max err {err:.3e} vs scale {scale:.3e} (rel {err / scale:.2e} > {rtol:.1e})"


def grad_close(name, got, ref, rtol=None):
    """gradient of parameter `name`: max-norm relative error.  The two tensors behind the scoring ReLU go through
    `scores_grads_close` (kink-aware, same tolerance) -- there is no looser bar for any tensor."""
    assert "to_scores." not in name, "to_scores gradients are compared by scores_grads_close / net_grads_close (kink-aware)"
    return maxnorm_close(got, ref, GRAD_RTOL if rtol is None else rtol, name)


KINK_BAND = 1e-5      # |z| < KINK_BAND * max|z|: a scoring pre-activation within rounding of the ReLU kink (SAST.py:110)
KINK_MAX_PER_ROW = 16 # ambiguous elements allowed per output channel: the 0/1 fit below (k <= 16 unknowns, C_in + 1 >= 65 equations at the sizes where k > 4 occurs) must stay heavily
                      # over-determined so that it cannot absorb a real error (full-size rows see 245 760 tokens: up to 9 in the band were measured)


def scores_grads_close(prefix, gW, gb, rW, rb, records, rtol=None):
    """d(to_scores.weight), d(to_scores.bias) of the scoring linear `prefix` against the reference's (rW, rb), kink-aware.

    scores = ReLU(z), z = x Ws^T + bs (SAST.py:110).  A (token, channel) element whose z lies within fp32 rounding of 0 is cut on one
    side and passes on the other, which moves row `channel` of dWs by that element's WHOLE contribution g * x (g = dL/dscores) --
    on the device exactly as between two CPU runs (tools/grad_error_probe.py --fp64).  `records` (oracle/sast_oracle.py: kink_log)
    hold x, z and g of every evaluation of this layer.  The elements with |z| < KINK_BAND * max|z| are taken OUT of the comparison on
    both sides: the reference loses the contributions of those it passed; the device's residual on a row must then be a 0/1
    combination of that row's ambiguous contributions (solved by least squares, rounded to {0, 1}) and is reduced by it.  What is
    left is compared at the ordinary gradient tolerance.
    The RAW, unadjusted error is capped too (round 6): element (c, i) of dWs may differ from the reference's by at most the sum of
    |g * x_i| over the ambiguous elements of row c (what flipping every one of them could move it by) plus the ordinary tolerance, and
    db likewise -- so the fit can only forgive what the kink itself can explain: a row without ambiguous elements sits at the plain
    tolerance, and an error larger than the ambiguous contributions fails even where the 0/1 fit could have absorbed part of it."""
    from conftest import record_error
    rtol = GRAD_RTOL if rtol is None else rtol
    gW, gb = gW.detach().double().cpu(), gb.detach().double().cpu()
    rW, rb = rW.detach().double().cpu(), rb.detach().double().cpu()
    scale_W, scale_b = float(rW.abs().max()) + 1e-12, float(rb.abs().max()) + 1e-12
    record_error(_test_id(), prefix + "to_scores.weight [raw, diagnostic]", float((gW - rW).abs().max()), scale_W, rtol)
    C = rW.shape[0]
    xs = torch.cat([r["x"].detach().reshape(-1, r["x"].shape[-1]) for r in records]).double()
    zs = torch.cat([r["z"].detach().reshape(-1, C) for r in records]).double()
    gs = torch.cat([(r["s"].grad if r["s"].grad is not None else torch.zeros_like(r["s"])).reshape(-1, C) for r in records]).double()
    amb = zs.abs() < KINK_BAND * float(zs.abs().max())
    n_amb = int(amb.sum())
    on = amb & (zs > 0)
    # raw cap: |device - reference| <= sum over the row's ambiguous elements of |g x| (+ the ordinary tolerance), element by element
    capW = (gs.abs() * amb).t() @ xs.abs()
    capb = (gs.abs() * amb).sum(0)
    rawW, rawb = (gW - rW).abs(), (gb - rb).abs()
    overW, overb = float((rawW - capW).max()), float((rawb - capb).max())
    record_error(_test_id(), prefix + "to_scores.weight [raw minus what the ambiguous elements can move]", max(overW, 0.0), scale_W, rtol)
    assert overW <= rtol * scale_W + 1e-9, (f"{prefix}to_scores.weight: raw error exceeds what the {n_amb} ambiguous pre-activations can explain by "
                                            f"{overW:.3e} (scale {scale_W:.3e}, rel {overW / scale_W:.2e} > {rtol:.1e})")
    assert overb <= rtol * scale_b + 1e-9, (f"{prefix}to_scores.bias: raw error exceeds what the ambiguous pre-activations can explain by "
                                            f"{overb:.3e} (scale {scale_b:.3e})")
    rW_off = rW - (gs * on).t() @ xs
    rb_off = rb - (gs * on).sum(0)
    gW_adj, gb_adj = gW.clone(), gb.clone()
    for c in torch.nonzero(amb.any(0)).view(-1).tolist():
        idx = torch.nonzero(amb[:, c]).view(-1)
        assert idx.numel() <= KINK_MAX_PER_ROW, (f"{prefix}to_scores row {c}: {idx.numel()} pre-activations inside the kink band "
                                                 f"(> {KINK_MAX_PER_ROW}): the kink-aware fit would have too much freedom -- use another seed")
        A = torch.cat([gs[idx, c, None] * xs[idx], gs[idx, c, None]], dim=1)            # (k, C_in + 1): contribution of each element
        res = torch.cat([gW[c] - rW_off[c], (gb[c] - rb_off[c]).view(1)])               # (C_in + 1,)
        if A.abs().max() == 0:
            continue
        m = torch.linalg.lstsq(A.t(), res[:, None]).solution.view(-1).clamp(0, 1).round()
        fix = m @ A
        gW_adj[c] -= fix[:-1]
        gb_adj[c] -= fix[-1]
    errW, errb = float((gW_adj - rW_off).abs().max()), float((gb_adj - rb_off).abs().max())
    record_error(_test_id(), prefix + f"to_scores.weight [kink-aware, {n_amb} ambiguous]", errW, scale_W, rtol)
    record_error(_test_id(), prefix + "to_scores.bias [kink-aware]", errb, scale_b, rtol)
    assert errW <= rtol * scale_W + 1e-9, f"{prefix}to_scores.weight: kink-aware max err {errW:.3e} vs scale {scale_W:.3e} ({n_amb} ambiguous elements)"
    assert errb <= rtol * scale_b + 1e-9, f"{prefix}to_scores.bias: kink-aware max err {errb:.3e} vs scale {scale_b:.3e} ({n_amb} ambiguous elements)"


def net_grads_close(named_params, ref_grad, kink_log, rtol=None, prefix="", log_prefix=""):
    """every parameter gradient of a module against the reference's; `ref_grad(name)` -> tensor or None (parameter unused);
    `kink_log`: the oracle's record of the scoring layers (keys = oracle prefixes = log_prefix + module prefix + '...att.')."""
    got = {k: v for k, v in named_params if "sub_layers" not in k}
    for k, v in got.items():
        ref = ref_grad(k)
        if ref is None:
            continue
        if k.endswith("to_scores.weight"):
            pre = k[:-len("to_scores.weight")]
            scores_grads_close(prefix + pre, v.grad, got[pre + "to_scores.bias"].grad, ref, ref_grad(pre + "to_scores.bias"),
                               kink_log[log_prefix + pre], rtol)
        elif k.endswith("to_scores.bias"):
            continue
        else:
            grad_close(prefix + k, v.grad, ref, rtol)
