"""Run-to-run reproducibility of one training step on the MI355X: the same step twice from identical
state, gradients compared per parameter tensor; then two 4-step training loops from the same initial state.

    python tools/determinism_probe.py [--full]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sast_amd.config import backbone_config            # noqa: E402
from sast_amd.detection import RNNDetector, YOLOPAFPN   # noqa: E402
from sast_amd.dist import FlatParams, FusedAdamW         # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="1Mpx B=4 instead of the 128x160 B=2 test configuration")
    ap.add_argument("--amp", type=float, default=2e-4)
    ap.add_argument("--ls", type=float, default=0.5)
    ap.add_argument("--eps", type=float, default=1e-8, help="AdamW eps (1e-8: an element whose gradient is rounding noise moves by +-lr)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    if args.full:
        hw, part, E, B, chans = (384, 640), (6, 10), 64, 4, (128, 256, 512)
    else:
        hw, part, E, B, chans = (128, 160), (4, 5), 32, 2, (64, 128, 256)
    g = torch.Generator().manual_seed(2)
    x = ((torch.rand(B, 20, hw[0], hw[1], generator=g) < 0.05) * torch.randint(1, 11, (B, 20, hw[0], hw[1]), generator=g)).to(torch.uint8).to(dev)

    def make():
        torch.manual_seed(0)
        net = RNNDetector(backbone_config(hw, part, embed_dim=E, AMP=args.amp, ls_init_value=args.ls)).to(dev)
        fpn = YOLOPAFPN(depth=0.67, in_stages=(2, 3, 4), in_channels=chans).to(dev)
        flat = FlatParams([net, fpn])
        names = {}
        for m, pre in ((net, "net."), (fpn, "fpn.")):
            for k, p in m.named_parameters():
                names.setdefault(id(p), pre + k)
        return net, fpn, flat, FusedAdamW(flat, lr=1e-3, eps=args.eps), [names[id(p)] for p in flat.params]

    def step(net, fpn, flat, opt, names, update):
        flat.zero_grad()
        feats, _s, P = net.forward_nhwc(x)
        outs = fpn.forward_nhwc(feats)
        loss = sum((o * o).mean() for o in outs)
        loss.backward()
        if update:
            opt.step()
        return float(loss), [int(p) for p in P], [o.detach().clone() for o in outs]

    def per_param(flat, names, ga, gb, top=8):
        rows = []
        for p, n, off in zip(flat.params, names, flat.offsets):
            k = p.numel()
            a, b = ga[off:off + k], gb[off:off + k]
            sc = float(b.abs().max())
            rows.append((float((a - b).abs().max()) / (sc + 1e-30), n, sc))
        rows.sort(reverse=True)
        return rows[:top]

    print(f"== config {'1Mpx B=4' if args.full else '128x160 B=2'}  AMP={args.amp} ls={args.ls}")
    m = make()
    res = []
    for i in range(3):
        l, P, outs = step(*m, update=False)
        res.append((l, P, outs, m[2].grad.clone()))
    for i in (1, 2):
        dg = float((res[i][3] - res[0][3]).abs().max()) / float(res[0][3].abs().max())
        do = max(float((a - b).abs().max()) for a, b in zip(res[i][2], res[0][2]))
        print(f"same state, run {i} vs 0: loss {res[i][0]!r} vs {res[0][0]!r}; P equal {res[i][1] == res[0][1]}; forward outputs max abs diff {do:.3e}; "
              f"grad max-norm rel diff {dg:.3e}")
        for r in per_param(m[2], m[4], res[i][3], res[0][3]):
            print(f"     {r[0]:.3e}  {r[1]}  (scale {r[2]:.3e})")
    runs = []
    for rep in range(2):
        m = make()
        seq = []
        for k in range(4):
            l, P, _o = step(*m, update=True)
            seq.append((l, P, m[2].grad.clone()))
        runs.append((seq, m))
    for k in range(4):
        a, b = runs[0][0][k], runs[1][0][k]
        dg = float((a[2] - b[2]).abs().max()) / float(b[2].abs().max())
        print(f"two training loops, step {k}: loss {a[0]!r} vs {b[0]!r}; P equal {a[1] == b[1]} {a[1]} ; grad max-norm rel diff {dg:.3e}")
        for r in per_param(runs[0][1][2], runs[0][1][4], a[2], b[2], top=4):
            print(f"     {r[0]:.3e}  {r[1]}  (scale {r[2]:.3e})")


if __name__ == "__main__":
    main()
