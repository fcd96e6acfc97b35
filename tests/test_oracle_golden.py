"""CPU: the oracle (oracle/sast_oracle.py) against the fixtures captured from the reference.

Fixtures were produced by tests/golden/make_golden.py by importing /root/reference in the
build container; parameters are re-drawn here from the recorded seed (checksum-guarded).
Bar: index lists exact; floating point torch.equal (same ATen op sequence) -- we allow
1e-6 in case a BLAS build differs between boxes.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import sast_oracle as O

import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))

ATOL = 1e-6
LIST_NAMES = ("index_window", "index_token", "padding_index", "asy_index", "K")


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)


def _block_params(C, seed, nblocks=1):
    cfg = O.BackboneCfg(in_res_hw=(64, 80), partition_size=(4, 5), embed_dim=C, num_blocks=(nblocks, 1, 1, 1))
    p = O.init_backbone_params(cfg, seed=seed, ls_init=0.5)
    return {k[len("stages.0."):]: v for k, v in p.items() if k.startswith("stages.0.att_blocks.")}


def _checksum(params):
    return float(sum(float(v.double().abs().sum()) for v in params.values()))


def test_non_zero_ratio(golden_dir):
    g = _load(golden_dir, "nzr")
    assert torch.equal(O.non_zero_ratio(torch.from_numpy(g["x"])), torch.from_numpy(g["r"]))
    assert torch.equal(O.non_zero_ratio(torch.from_numpy(g["xb"])), torch.from_numpy(g["rb"]))


@pytest.mark.parametrize("name", ["block_amp2e-4", "block_amp2e-2", "block_amp1", "block_b1", "block_cb", "block_small_dh24",
                                  "block_large_c96", "block_nobias", "block_act_relu", "block_act_silu", "block_act_sigmoid",
                                  "block_act_tanh", "block_act_mish", "block_act_relu6", "block_act_leaky_relu", "block_act_elu", "block_act_celu",
                                  "block_act_selu", "block_act_hard_sigmoid", "block_act_hard_swish", "block_act_hard_mish", "block_dh16", "block_dh8",
                                  "block_t240_dense", "block_t240_sparse", "block_act_prelu"])
def test_sast_block(golden_dir, name):
    g = _load(golden_dir, name)
    x, r = torch.from_numpy(g["x"]), torch.from_numpy(g["r"])
    params = _block_params(x.shape[-1], int(g["seed"]))
    if "bias" in g and not int(g["bias"]):       # attention_bias: False, mlp_bias: False -- the linears have no bias vectors
        params = {k: v for k, v in params.items() if not (k.endswith(".bias") and ("qkv." in k or "proj." in k or "mlp.net" in k))}
    if "prelu_slopes" in g:                      # mlp_activation prelu: one learnable slope per layer (layers/activations.py:124-131)
        for layer, slope in zip(("win_attn", "grid_attn"), g["prelu_slopes"]):
            params[f"att_blocks.0.att.{layer}.mlp.net.0.act_layer.weight"] = torch.tensor([float(slope)])
    assert abs(_checksum(params) - float(g["param_checksum"])) < 1e-6 * float(g["param_checksum"])
    part = tuple(int(v) for v in g["part"]) if "part" in g else (4, 5)    # block_t240_*: partitions of 12 x 20 = 240 tokens
    cfg = O.AttnCfg(partition_size=part, amp=float(g["amp"]), bounce=1e-3, enable_cb=bool(g["enable_cb"]),
                    dim_head=int(g["dim_head"]) if "dim_head" in g else 32,
                    mlp_activation=str(g["act"]) if "act" in g else "gelu")     # mlp_activation: relu / silu / sigmoid / tanh fixtures
    pe = O.position_embedding_sine(x.shape[1], x.shape[2], x.shape[3])
    xo = x.clone().requires_grad_(True)
    po = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    out, cnt, lists = O.sast_block(xo, pe, r, po, "att_blocks.0.att.", cfg)
    assert cnt == int(g["count"])
    for li, l in enumerate(lists):
        for nm, t in zip(LIST_NAMES, l):
            assert np.array_equal(t.numpy(), g[f"l{li}_{nm}"]), (li, nm)
    assert torch.allclose(out, torch.from_numpy(g["out"]), atol=ATOL, rtol=0)
    (out ** 2).mean().backward()
    assert torch.allclose(xo.grad, torch.from_numpy(g["dx"]), atol=1e-7, rtol=1e-4)
    for k, v in po.items():
        gk = "g_" + k[len("att_blocks.0.att."):]
        assert torch.allclose(v.grad, torch.from_numpy(g[gk]), atol=1e-7, rtol=1e-3), k


@pytest.mark.parametrize("name", ["block_drop_path", "block_drop_mlp", "block_drop_path_cb"])
def test_sast_block_drop_path(golden_dir, name):
    """drop_path > 0 (SAST.py:42,188,193,232,248): the reference block in training mode under a fixed RNG state.  The oracle reproduces
    outputs, index lists and every gradient both with the four recorded factor vectors injected and by drawing them itself from the same RNG
    state; eval mode ignores DropPath."""
    g = _load(golden_dir, name)
    x, r = torch.from_numpy(g["x"]), torch.from_numpy(g["r"])
    params = _block_params(x.shape[-1], int(g["seed"]))
    assert abs(_checksum(params) - float(g["param_checksum"])) < 1e-6 * float(g["param_checksum"])
    pmlp = float(g["p_mlp"])           # block_drop_mlp: `drop_mlp` (nn.Dropout on the MLP hidden, ops.py:167) instead of DropPath
    masks = [torch.from_numpy(g[k]) for k in sorted((k for k in g.files if k.startswith("drop")), key=lambda k: int(k[4:]))]
    pdp, cb = float(g["p"]), bool(int(g["enable_cb"])) if "enable_cb" in g else False    # block_drop_path_cb: both dropouts + Context Broadcasting
    want = []      # call order per layer (SAST.py:232-248): DropPath factors (kept rows), MLP mask (kept rows x inner), DropPath factors
    for li in (0, 1):
        k = len(g[f"l{li}_asy_index"])
        want += ([(k,)] if pdp else []) + ([(k, 64)] if pmlp else []) + ([(k,)] if pdp else [])
    assert [tuple(m.shape) for m in masks] == want
    for m in masks:
        keep = 1.0 - (pmlp if m.dim() == 2 else pdp)
        assert set(torch.unique(m).tolist()) <= {0.0, float(np.float32(1.0) / np.float32(keep))}
    pe = O.position_embedding_sine(x.shape[1], x.shape[2], x.shape[3])
    for mode in ("inject", "draw"):
        cfg = O.AttnCfg(partition_size=(4, 5), amp=float(g["amp"]), bounce=1e-3, drop_path=pdp, drop_mlp=pmlp, training=True, enable_cb=cb,
                        drop_masks=[m.clone() for m in masks] if mode == "inject" else None)
        xo = x.clone().requires_grad_(True)
        po = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        torch.manual_seed(int(g["rng_seed"]))
        out, cnt, lists = O.sast_block(xo, pe, r, po, "att_blocks.0.att.", cfg)
        assert cnt == int(g["count"])
        for li, l in enumerate(lists):
            for nm, t in zip(LIST_NAMES, l):
                assert np.array_equal(t.numpy(), g[f"l{li}_{nm}"]), (li, nm)
        assert torch.allclose(out, torch.from_numpy(g["out"]), atol=ATOL, rtol=0), mode
        (out ** 2).mean().backward()
        assert torch.allclose(xo.grad, torch.from_numpy(g["dx"]), atol=1e-7, rtol=1e-4)
        for k, v in po.items():
            assert torch.allclose(v.grad, torch.from_numpy(g["g_" + k[len("att_blocks.0.att."):]]), atol=1e-7, rtol=1e-3), (mode, k)
    ev, _c, _l = O.sast_block(x, pe, r, params, "att_blocks.0.att.", O.AttnCfg(partition_size=(4, 5), amp=float(g["amp"]), drop_path=pdp,
                                                                              drop_mlp=pmlp, training=False, enable_cb=cb))
    assert torch.allclose(ev, torch.from_numpy(g["eval_out"]), atol=ATOL, rtol=0)


def test_downsample_no_overlap_no_affine(golden_dir):
    """downsample_cfg.overlap False / norm_affine False (ops.py:69-76,87): the oracle against the reference module's numbers"""
    g = _load(golden_dir, "downsample_variants")
    for tag, f in (("f4", 4), ("f2", 2)):
        x = torch.from_numpy(g[tag + "_x"]).requires_grad_(True)
        p = {"conv.weight": torch.from_numpy(g[tag + "_w"]).requires_grad_(True)}
        assert p["conv.weight"].shape[-1] == f
        y = O.conv_downsample_cf2cl(x, p, "", f)
        assert torch.allclose(y, torch.from_numpy(g[tag + "_y"]), atol=1e-6, rtol=0)
        (y * torch.from_numpy(g[tag + "_wy"])).sum().backward()
        assert torch.allclose(x.grad, torch.from_numpy(g[tag + "_dx"]), atol=1e-6, rtol=1e-4)
        assert torch.allclose(p["conv.weight"].grad, torch.from_numpy(g[tag + "_dw"]), atol=1e-5, rtol=1e-4)


def test_two_blocks_reuse_index_lists(golden_dir):
    g = _load(golden_dir, "stage_two_blocks")
    x, r = torch.from_numpy(g["x"]), torch.from_numpy(g["r"])
    params = _block_params(64, int(g["seed"]), nblocks=2)
    cfg = O.AttnCfg(partition_size=(4, 5), amp=float(g["amp"]))
    pe = O.position_embedding_sine(16, 20, 64)
    a1, c1, l1 = O.sast_block(x, pe, r, params, "att_blocks.0.att.", cfg)
    a2, c2, _ = O.sast_block(a1, pe, r, params, "att_blocks.1.att.", cfg, index_list=l1, first_block=False)
    assert (c1, c2) == (int(g["count1"]), int(g["count2"]))
    assert torch.allclose(a1, torch.from_numpy(g["out1"]), atol=ATOL, rtol=0)
    assert torch.allclose(a2, torch.from_numpy(g["out2"]), atol=ATOL, rtol=0)


@pytest.mark.parametrize("tag", ["dense", "sparse"])
def test_backbone_tiny(golden_dir, tag):
    g = _load(golden_dir, f"backbone_tiny_{tag}")
    cfg = O.BackboneCfg(in_res_hw=(128, 160), partition_size=(4, 5), embed_dim=32, amp=float(g["amp"]))
    params = O.init_backbone_params(cfg, seed=int(g["seed"]), ls_init=0.5)
    assert abs(_checksum(params) - float(g["param_checksum"])) < 1e-6 * float(g["param_checksum"])
    po = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    o0, s0, P0 = O.backbone(torch.from_numpy(g["x0"]), None, po, cfg)
    o1, s1, P1 = O.backbone(torch.from_numpy(g["x1"]), [(h.detach(), c.detach()) for h, c in s0], po, cfg)
    assert P0 == list(g["P0"]) and P1 == list(g["P1"])
    for k in (1, 2, 3, 4):
        assert torch.allclose(o0[k].detach(), torch.from_numpy(g[f"h0_{k}"]), atol=ATOL, rtol=0)
        assert torch.allclose(o1[k].detach(), torch.from_numpy(g[f"h1_{k}"]), atol=ATOL, rtol=0)
        assert torch.allclose(s1[k - 1][1].detach(), torch.from_numpy(g[f"c1_{k}"]), atol=ATOL, rtol=0)
    loss = sum((o1[k] ** 2).mean() for k in (1, 2, 3, 4))
    assert abs(float(loss) - float(g["loss"])) < 1e-6 * abs(float(g["loss"]))
    loss.backward()
    stats = json.loads(str(g["grad_stats_json"]))
    for k, (nrm, _sm) in stats.items():
        got = float(po[k].grad.double().norm())
        assert abs(got - nrm) <= 1e-4 * nrm + 1e-10, k
    for key in g.files:
        if key.startswith("g_"):
            assert torch.allclose(po[key[2:]].grad, torch.from_numpy(g[key]), atol=1e-7, rtol=1e-3), key


def test_pafpn(golden_dir):
    g = _load(golden_dir, "pafpn")
    chans = (64, 128, 256)
    params = O.init_pafpn_params(chans, seed=int(g["seed"]))
    assert abs(_checksum(params) - float(g["param_checksum"])) < 1e-6 * float(g["param_checksum"])
    feats = {k: torch.from_numpy(g[f"in{k}"]).requires_grad_(True) for k in (2, 3, 4)}
    po = {k: (v.clone().requires_grad_(True) if "running" not in k else v.clone()) for k, v in params.items()}
    outs = O.pafpn(feats, po, training=True, bufs=po)
    for i, o in enumerate(outs):
        assert torch.allclose(o.detach(), torch.from_numpy(g[f"train_out{i}"]), atol=ATOL, rtol=0)
    assert torch.allclose(po["lateral_conv0.bn.running_mean"], torch.from_numpy(g["rm_lateral"]), atol=1e-7)
    assert torch.allclose(po["lateral_conv0.bn.running_var"], torch.from_numpy(g["rv_lateral"]), atol=1e-7)
    sum((o ** 2).mean() for o in outs).backward()
    for k in (2, 3, 4):
        assert torch.allclose(feats[k].grad, torch.from_numpy(g[f"din{k}"]), atol=1e-7, rtol=1e-3)
    stats = json.loads(str(g["grad_stats_json"]))
    for k, (nrm, _sm) in stats.items():
        got = float(po[k].grad.double().norm())
        assert abs(got - nrm) <= 1e-4 * nrm + 1e-10, k
    with torch.no_grad():
        ev = O.pafpn({k: v.detach() for k, v in feats.items()}, po, training=False)
    for i, o in enumerate(ev):
        assert torch.allclose(o, torch.from_numpy(g[f"eval_out{i}"]), atol=ATOL, rtol=0)


def test_lstm_cell_update_dropout(golden_dir):
    """cell_update_dropout > 0 (rnn.py:34,64): the fixture holds the reference DWSConvLSTM2d's training-mode numbers under a fixed RNG state
    and the keep mask it drew.  The oracle with that mask, and the oracle drawing its own under the same RNG state (the same nn.Dropout call
    on a tensor of the same shape), both reproduce them; eval mode ignores the dropout."""
    g = _load(golden_dir, "lstm_dropout")
    cfgp = O.BackboneCfg(in_res_hw=(128, 160), partition_size=(4, 5), embed_dim=32)
    full = O.init_backbone_params(cfgp, seed=int(g["seed"]))
    params = {k[len("stages.0."):]: v for k, v in full.items() if k.startswith("stages.0.lstm.")}
    x, h0, c0, mask = (torch.from_numpy(g[k]) for k in ("x", "h0", "c0", "mask"))
    wh, wc, pdrop = torch.from_numpy(g["wh"]), torch.from_numpy(g["wc"]), float(g["p"])
    assert set(torch.unique(mask).tolist()) == {0.0, float(np.float32(1.0 / (1.0 - pdrop)))}
    po = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    xo, ho, co = x.clone().requires_grad_(True), h0.clone().requires_grad_(True), c0.clone().requires_grad_(True)
    h1, c1 = O.conv_lstm(xo, (ho, co), po, "lstm.", drop_mask=mask)
    assert torch.allclose(h1, torch.from_numpy(g["h1"]), atol=1e-6, rtol=0) and torch.allclose(c1, torch.from_numpy(g["c1"]), atol=1e-6, rtol=0)
    ((h1 * wh).sum() + (c1 * wc).sum()).backward()
    for got, key in ((xo.grad, "dx"), (ho.grad, "dh0"), (co.grad, "dc0")):
        assert torch.allclose(got, torch.from_numpy(g[key]), atol=1e-6, rtol=1e-4), key
    for k, v in po.items():
        assert torch.allclose(v.grad, torch.from_numpy(g["g_" + k[len("lstm."):]]), atol=1e-5, rtol=1e-4), k
    torch.manual_seed(int(g["rng_seed"]))
    h1b, c1b = O.conv_lstm(x, (h0, c0), params, "lstm.", cell_update_dropout=pdrop, training=True)
    assert torch.equal(h1b, h1.detach()) and torch.equal(c1b, c1.detach())
    he, ce = O.conv_lstm(x, (h0, c0), params, "lstm.", cell_update_dropout=pdrop, training=False)
    assert torch.allclose(he, torch.from_numpy(g["eval_h1"]), atol=1e-6, rtol=0) and torch.allclose(ce, torch.from_numpy(g["eval_c1"]), atol=1e-6, rtol=0)


def test_depthwise_pafpn_and_head(golden_dir):
    """depthwise=True (yolo_pafpn.py:37, network_blocks.py:57-76,93, yolo_head.py:42): the oracle's DWConv units against the reference
    modules' numbers -- PAFPN train mode (outputs, running statistics of a depth-wise BatchNorm, input gradients, every parameter gradient
    norm, four gradient tensors), PAFPN eval mode, head eval mode on the PAFPN's outputs."""
    g = _load(golden_dir, "depthwise")
    chans = (32, 64, 128)
    params = O.init_pafpn_params(chans, seed=int(g["seed"]), depthwise=True)
    assert abs(_checksum(params) - float(g["param_checksum"])) < 1e-6 * float(g["param_checksum"])
    assert params["bu_conv2.dconv.conv.weight"].shape == (32, 1, 3, 3) and "bu_conv2.conv.weight" not in params
    feats = {k: torch.from_numpy(g[f"in{k}"]).requires_grad_(True) for k in (2, 3, 4)}
    po = {k: (v.clone().requires_grad_(True) if "running" not in k else v.clone()) for k, v in params.items()}
    outs = O.pafpn(feats, po, training=True, bufs=po)
    for i, o in enumerate(outs):
        assert torch.allclose(o.detach(), torch.from_numpy(g[f"train_out{i}"]), atol=ATOL, rtol=0)
    assert torch.allclose(po["bu_conv2.dconv.bn.running_mean"], torch.from_numpy(g["rm_bu_dconv"]), atol=1e-7)
    assert torch.allclose(po["bu_conv2.dconv.bn.running_var"], torch.from_numpy(g["rv_bu_dconv"]), atol=1e-7)
    sum((o ** 2).mean() for o in outs).backward()
    for k in (2, 3, 4):
        assert torch.allclose(feats[k].grad, torch.from_numpy(g[f"din{k}"]), atol=1e-7, rtol=1e-3)
    for k, (nrm, _sm) in json.loads(str(g["grad_stats_json"])).items():
        assert abs(float(po[k].grad.double().norm()) - nrm) <= 1e-4 * nrm + 1e-10, k
    for k in ("bu_conv2.dconv.conv.weight", "bu_conv2.pconv.conv.weight", "C3_p3.m.0.conv2.dconv.conv.weight", "bu_conv1.dconv.bn.weight"):
        assert torch.allclose(po[k].grad, torch.from_numpy(g["g_" + k]), atol=1e-7, rtol=1e-3), k
    with torch.no_grad():
        ev = O.pafpn({k: v.detach() for k, v in feats.items()}, po, training=False)
    for i, o in enumerate(ev):
        assert torch.allclose(o, torch.from_numpy(g[f"eval_out{i}"]), atol=ATOL, rtol=0)
    hp = O.init_head_params(chans, num_classes=int(g["num_classes"]), seed=int(g["head_seed"]), depthwise=True)
    hout = O.yolox_head_eval([torch.from_numpy(g[f"eval_out{i}"]) for i in range(3)], hp)
    assert torch.allclose(hout, torch.from_numpy(g["head_out"]), atol=1e-5, rtol=1e-5)


def test_yolox_head_eval(golden_dir):
    """YOLOX head inference path: oracle restatement vs the reference module's outputs (decoded and raw)."""
    g = _load(golden_dir, "head_eval")
    chans = (64, 128, 256)
    params = O.init_head_params(chans, num_classes=int(g["num_classes"]), seed=int(g["seed"]))
    feats = [torch.from_numpy(g[f"in{i}"]) for i in range(3)]
    assert torch.allclose(O.yolox_head_eval(feats, params), torch.from_numpy(g["out"]), atol=1e-6, rtol=1e-6)
    assert torch.allclose(O.yolox_head_eval(feats, params, decode=False), torch.from_numpy(g["raw"]), atol=1e-6, rtol=1e-6)


def test_backbone_masked(golden_dir):
    """enable_masking (sast_rnn.py:271-273): oracle vs the reference backbone run with a token mask."""
    g = _load(golden_dir, "backbone_masked")
    hw, part, E = (128, 160), (4, 5), 32
    ocfg = O.BackboneCfg(in_res_hw=hw, partition_size=part, embed_dim=E, amp=2e-2)
    params = O.init_backbone_params(ocfg, seed=int(g["seed"]), ls_init=0.5)
    params["stages.0.mask_token"] = torch.from_numpy(g["mask_token"])
    po = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    out, _s, P = O.backbone(torch.from_numpy(g["x"]), None, po, ocfg, token_mask=torch.from_numpy(g["mask"]).bool())
    assert [int(p) for p in P] == list(g["P"])
    for k in (1, 2, 3, 4):
        assert torch.allclose(out[k], torch.from_numpy(g[f"h{k}"]), atol=ATOL, rtol=0)
    sum((out[k] ** 2).mean() for k in (1, 2, 3, 4)).backward()
    assert torch.allclose(po["stages.0.mask_token"].grad, torch.from_numpy(g["g_mask_token"]), atol=1e-8, rtol=1e-4)


def test_postprocess_known_answer():
    """greedy class-aware NMS of the oracle on a hand-checked case (the reference's torchvision call cannot run here)."""
    # (cx, cy, w, h, obj, cls0, cls1): boxes 0/1 overlap heavily (IoU 0.68) and share class 0 -> 1 suppressed; box 2 overlaps 0 but is
    # class 1 -> kept; box 3 is below the confidence threshold; box 4 is far away
    p = torch.tensor([[[50., 50., 40., 40., 0.9, 0.9, 0.1],
                       [54., 50., 40., 40., 0.8, 0.9, 0.1],
                       [52., 50., 40., 40., 0.7, 0.1, 0.9],
                       [50., 50., 40., 40., 0.2, 0.5, 0.1],
                       [200., 200., 30., 30., 0.6, 0.8, 0.2]]])
    out = O.postprocess(p, 2, conf_thre=0.3, nms_thre=0.45)[0]
    assert out.shape == (3, 7)
    assert torch.allclose(out[:, 4] * out[:, 5], torch.tensor([0.81, 0.63, 0.48]), atol=1e-6)
    assert out[:, 6].tolist() == [0.0, 1.0, 0.0]
    assert torch.allclose(out[0, :4], torch.tensor([30., 30., 70., 70.]))


def test_yolox_head_train(golden_dir):
    """YOLOX training branch (SimOTA assignment, IoU / objectness / class losses): oracle vs the reference module's numbers."""
    g = _load(golden_dir, "head_train")
    chans, nc = (64, 128, 256), int(g["num_classes"])
    params = {k: (v.clone().requires_grad_(True) if "running" not in k else v.clone())
              for k, v in O.init_head_params(chans, num_classes=nc, seed=int(g["seed"])).items()}
    feats = [torch.from_numpy(g[f"in{i}"]).requires_grad_(True) for i in range(3)]
    r = O.yolox_head_train(feats, torch.from_numpy(g["labels"]), params, num_classes=nc)
    for k in ("loss", "iou_loss", "conf_loss", "cls_loss", "num_fg"):
        assert abs(float(r[k]) - float(g[k])) <= 1e-5 * max(1.0, abs(float(g[k]))), k
    for b, (fg, matched, pious) in enumerate(r["assign"]):
        assert np.array_equal(fg.numpy().astype(np.uint8), g[f"fg{b}"]) and np.array_equal(matched.numpy(), g[f"matched{b}"])
        assert np.allclose(pious.numpy(), g[f"piou{b}"], atol=1e-6)
    r["loss"].backward()
    for i, f in enumerate(feats):
        assert torch.allclose(f.grad, torch.from_numpy(g[f"din{i}"]), atol=1e-7, rtol=1e-4)


def test_full_size_stats_g1(golden_dir):
    """F-7 (Gen1 size; the 1Mpx twin runs on the GPU box next to the HIP path)."""
    with open(os.path.join(golden_dir, "full_stats.json")) as f:
        ref = json.load(f)["G1"]
    cfg = O.BackboneCfg(in_res_hw=(256, 320), partition_size=(8, 10))
    params = O.init_backbone_params(cfg, seed=0)
    x = O.synthetic_events(4, (256, 320), seed=0, sparsity=0.9)
    with torch.no_grad():
        out, _st, P, lists = O.backbone(x, None, params, cfg, return_lists=True)
    assert P == ref["P"]
    assert [[len(l[3]) for l in ls[0]] for ls in lists] == ref["sumK"]
    for k in (1, 2, 3, 4):
        t = out[k].double()
        assert abs(float(t.abs().mean()) - ref[f"h{k}"]["absmean"]) < 1e-6


@pytest.mark.skipif(not os.path.isdir("/root/reference/models"), reason="reference only exists in the build container")
def test_oracle_equals_reference_live():
    """container-only: run the imported reference next to the oracle on a fresh seed."""
    import _ref_import as RI
    ref = RI.import_reference()
    hw, part = (128, 160), (4, 5)
    rcfg = RI.backbone_cfg(hw, part, embed_dim=32, amp=2e-3, ls_init=0.5)
    cfg = O.BackboneCfg(in_res_hw=hw, partition_size=part, embed_dim=32, amp=2e-3)
    params = O.init_backbone_params(cfg, seed=5, ls_init=0.5)
    net = ref.sast_rnn.RNNDetector(rcfg)
    from make_golden import load_into
    load_into(net, params)
    x = O.count_events(3, hw, seed=9, density=0.03)
    with torch.no_grad():
        a, _, Pa = net(x)
        b, _, Pb = O.backbone(x, None, params, cfg)
    assert Pa == Pb
    for k in (1, 2, 3, 4):
        assert torch.equal(a[k], b[k])


def test_sequence_gather_and_rnn_states(golden_dir):
    """(f)2: oracle restatement of BackboneFeatureSelector / RNNStates.reset against the imported reference's outputs"""
    import json
    g = _load(golden_dir, "sequence_gather")
    idx_seq = json.loads(str(g["idx_json"]))
    feats_seq = [{k: torch.from_numpy(g[f"f{t}_{k}"]) for k in (1, 2, 3, 4)} for t in range(len(idx_seq))]
    out = O.select_backbone_features(feats_seq, idx_seq)
    for k in (1, 2, 3, 4):
        assert torch.equal(out[k], torch.from_numpy(g[f"out_{k}"]))
    states = [(torch.from_numpy(g[f"h_{i}"]), torch.from_numpy(g[f"c_{i}"])) for i in range(4)]
    rs = O.rnn_states_reset(states, torch.tensor([True, False, True]))
    for i in range(4):
        assert torch.equal(rs[i][0], torch.from_numpy(g[f"reset_bool_h_{i}"])) and torch.equal(rs[i][1], torch.from_numpy(g[f"reset_bool_c_{i}"]))
    assert O.select_backbone_features(feats_seq, [[], None, [], []]) is None


def _sparse_cases(golden_dir=None):
    import os as _os
    d = golden_dir or _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden")
    with open(_os.path.join(d, "full_stats_sparse.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("key", sorted(_sparse_cases().keys()))
def test_full_size_sparse_selection_of_the_reference(golden_dir, key):
    """F-7, sparse: the oracle against what the IMPORTED REFERENCE selected at full size with less than half of the tokens kept
    (tests/golden/full_stats_sparse.json, written by make_golden.py --sparse-only): sha256 of index_window / asy_index / K of every
    stage and layer, P, M, sum K, output statistics.  The same hashes are asserted on the HIP path by
    tests/test_gpu_parity.py::test_full_size_sparse_selection_vs_reference."""
    import hashlib
    ref = _sparse_cases(golden_dir)[key]
    hw, part = tuple(ref["hw"]), tuple(ref["partition"])
    cfg = O.BackboneCfg(in_res_hw=hw, partition_size=part, amp=ref["amp"])
    params = O.init_backbone_params(cfg, seed=ref["seed"], ls_init=ref["ls_init"])
    x = O.count_events(ref["B"], hw, seed=100 + ref["seed"], density=ref["density"])
    with torch.no_grad():
        out, _st, P, lists = O.backbone(x, None, params, cfg, return_lists=True)
    assert [int(p) for p in P] == ref["P"]
    assert max(ref["kept_fraction"][:3]) < 0.5                     # really sparse
    for nm, idx in (("index_window", 0), ("asy_index", 3), ("K", 4)):
        got = [[hashlib.sha256(l[idx].numpy().astype(np.int64).tobytes()).hexdigest() for l in ls[0]] for ls in lists]
        assert got == ref[nm + "_sha256"], nm
    assert [[len(l[3]) for l in ls[0]] for ls in lists] == ref["sumK"]
    for k in (1, 2, 3, 4):
        t = out[k].double()
        assert abs(float(t.abs().mean()) - ref[f"h{k}"]["absmean"]) < 1e-6


@pytest.mark.parametrize("mode", ["hidden", "xh"])
def test_conv_lstm_depthwise(golden_dir, mode):
    """a12 with dws_conv=True (the reference class default): the oracle's conv_lstm against the reference DWSConvLSTM2d (fixture
    lstm_dws.npz) -- depth-wise conv on the previous hidden state / on cat(x, h), with a previous state and from the zero state."""
    g = _load(golden_dir, "lstm_dws")
    C = g["x"].shape[1]
    full = O.init_backbone_params(O.BackboneCfg(in_res_hw=(128, 160), partition_size=(4, 5), embed_dim=C), seed=int(g["seed"]), dws_conv=mode)
    params = {k[len("stages.0."):]: v for k, v in full.items() if k.startswith("stages.0.lstm.")}
    assert abs(_checksum(params) - float(g[mode + "_param_checksum"])) < 1e-6 * float(g[mode + "_param_checksum"])
    wh, wc = torch.from_numpy(g["wh"]), torch.from_numpy(g["wc"])
    for tag in ("prev", "zero"):
        x = torch.from_numpy(g["x"]).clone().requires_grad_(True)
        h0, c0 = torch.from_numpy(g["h0"]).clone().requires_grad_(True), torch.from_numpy(g["c0"]).clone().requires_grad_(True)
        po = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        h1, c1 = O.conv_lstm(x, (h0, c0) if tag == "prev" else None, po, "lstm.")
        pre = f"{mode}_{tag}_"
        assert torch.allclose(h1, torch.from_numpy(g[pre + "h1"]), atol=ATOL, rtol=0) and torch.allclose(c1, torch.from_numpy(g[pre + "c1"]), atol=ATOL, rtol=0)
        ((h1 * wh).sum() + (c1 * wc).sum()).backward()
        assert torch.allclose(x.grad, torch.from_numpy(g[pre + "dx"]), atol=1e-6, rtol=1e-4)
        for k, v in po.items():
            assert torch.allclose(v.grad, torch.from_numpy(g[pre + "g_" + k[len("lstm."):]]), atol=1e-5, rtol=1e-4), (tag, k)


def test_oracle_batched_nms_forms_differ_on_threshold_pairs():
    """the oracle restates torchvision.ops.batched_nms with its coordinate trick (oracle/sast_oracle.py:_batched_nms); on detections whose
    pairs sit on the IoU threshold it keeps a different set than the per-class evaluation of the unshifted boxes -- the case the GPU test
    test_postprocess_nms_coordinate_trick pins the kernel to"""
    from test_gpu_parity import _near_threshold_detections
    pred = _near_threshold_detections()
    out = O.postprocess(pred, 3, conf_thre=0.5, nms_thre=0.45)
    ip = pred[0]
    det = torch.cat((ip[:, 0:1] - ip[:, 2:3] / 2, ip[:, 1:2] - ip[:, 3:4] / 2, ip[:, 0:1] + ip[:, 2:3] / 2, ip[:, 1:2] + ip[:, 3:4] / 2), 1)
    cc, cp = torch.max(ip[:, 5:8], 1)
    assert len(O._nms_greedy(det, ip[:, 4] * cc, cp.float(), 0.45, False)) == 17 and out[0].shape[0] == 15
    # torchvision.ops.nms (class_agnostic) does not shift: at most what the per-class evaluation keeps
    assert O.postprocess(pred, 3, conf_thre=0.5, nms_thre=0.45, class_agnostic=True)[0].shape[0] <= 17
