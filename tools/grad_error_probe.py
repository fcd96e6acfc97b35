"""Per-parameter gradient error of the HIP path against the oracle at full size (1Mpx), to audit tolerances.

    python tools/grad_error_probe.py --batch 8 --amp 2e-2 --seed 1 [--fp64]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import sast_oracle as O                       # noqa: E402  (diagnostic tool, not the product path)
from test_gpu_parity import load_params, _rcfg, _cpu_lists   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--amp", type=float, default=2e-2)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--fp64", action="store_true", help="oracle in double (arbiter)")
    ap.add_argument("--no-fpn", action="store_true")
    ap.add_argument("--part", type=int, nargs=2, default=(6, 10), help="partition size: 6 10 (1Mpx default) / 12 20 (partition_split_32 1)")
    a = ap.parse_args()
    from sast_amd.detection import RNNDetector, YOLOPAFPN
    dev = torch.device("cuda:0")
    hw, part = (384, 640), tuple(a.part)
    ocfg = O.BackboneCfg(in_res_hw=hw, partition_size=part, amp=a.amp)
    params = O.init_backbone_params(ocfg, seed=a.seed, ls_init=0.5)
    fparams = O.init_pafpn_params((128, 256, 512), seed=a.seed + 50)
    net = RNNDetector(_rcfg(hw, part, 64, a.amp, 0.5)).to(dev)
    fpn = YOLOPAFPN(depth=0.67, in_stages=(2, 3, 4), in_channels=(128, 256, 512)).to(dev).train()
    load_params(net, params)
    load_params(fpn, fparams)
    x = O.count_events(a.batch, hw, seed=100 + a.seed, density=0.1)
    out, _st, P = net(x.to(dev))
    loss = 0.25 * sum((out[k] ** 2).mean() for k in (1, 2, 3, 4))
    if not a.no_fpn:
        outs = fpn({k: out[k] for k in (2, 3, 4)})
        loss = loss + sum((o ** 2).mean() for o in outs)
    loss.backward()
    lists = _cpu_lists(net)
    dt = torch.float64 if a.fp64 else torch.float32
    po = {k: v.clone().to(dt).requires_grad_(True) for k, v in params.items()}
    pf = {k: (v.clone().to(dt).requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in fparams.items()}
    if a.fp64:
        pf = {k: (v.double() if v.is_floating_point() and not v.requires_grad else v) for k, v in pf.items()}
    log = []
    oo, _s, Po = O.backbone(x.to(dt) if a.fp64 else x, None, po, ocfg, forced_lists=lists, diff_log=log)
    loss_o = 0.25 * sum((oo[k] ** 2).mean() for k in (1, 2, 3, 4))
    if not a.no_fpn:
        oouts = O.pafpn({k: oo[k] for k in (2, 3, 4)}, pf, training=True)
        loss_o = loss_o + sum((o ** 2).mean() for o in oouts)
    loss_o.backward()
    print("selection diffs:", [(d["win_diff"], d["tok_diff"], f"{d['max_margin']:.1e}") for d in log])
    print("loss", float(loss), float(loss_o), "P", [int(p) for p in P])
    for k in (1, 2, 3, 4):
        print(f"h{k} max abs err {float((out[k].detach().cpu().to(dt) - oo[k].detach()).abs().max()):.3e}")
    rows = []
    named = [(k, v) for k, v in net.named_parameters() if "sub_layers" not in k]
    ref = dict(po)
    if not a.no_fpn:
        named += [("fpn." + k, v) for k, v in fpn.named_parameters()]
        ref.update({"fpn." + k: v for k, v in pf.items()})
    for k, v in named:
        g, r = v.grad.detach().cpu().to(dt), ref[k].grad
        d = (g - r).abs()
        sc = float(r.abs().max()) + 1e-30
        rows.append((float(d.max()) / sc, float(d.pow(2).mean().sqrt()) / sc, k, sc, int(d.argmax()), tuple(r.shape)))
    rows.sort(reverse=True)
    print("rel max err   rel rms err   scale      argmax   shape   name")
    for r in rows[:25]:
        print(f"{r[0]:.3e}   {r[1]:.3e}   {r[3]:.2e}   {r[4]:8d} {str(r[5]):18s} {r[2]}")
    print("median rel max err over all tensors:", sorted(r[0] for r in rows)[len(rows) // 2])


if __name__ == "__main__":
    main()
