"""The fused MS-WSA layer kernels (csrc/k_mswsa_fused.hip) against the unfused launch chain (k_block.hip) on the same inputs, stage-1
shape of the 1Mpx model (96 x 160 map, C = 64, partitions 6 x 10), window and grid mode, dense and sparse selections: max-abs
difference of the outputs (and of every gradient with --bwd), and the time of both forms (graph-free, HIP events)."""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sast_amd import functional as SF

ap = argparse.ArgumentParser()
ap.add_argument("--bwd", action="store_true")
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--quick", action="store_true", help="one dense and one sparse window-mode case, no timing loop (used by the opt-in test)")
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, H, W, C, ph, pw, inner = args.batch, 96, 160, 64, 6, 10, 160
g = torch.Generator(device="cpu").manual_seed(1)


def params():
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    p = dict(ln1_w=1 + r(C, sc=0.1), ln1_b=r(C, sc=0.1), ln2_w=1 + r(C, sc=0.1), ln2_b=r(C, sc=0.1), qkv_w=r(3 * C, C, sc=C ** -0.5), qkv_b=r(3 * C, sc=0.1),
             proj_w=r(C, C, sc=C ** -0.5), proj_b=r(C, sc=0.1), ls1=0.5 + r(C, sc=0.1), fc1_w=r(2 * inner, C, sc=C ** -0.5), fc1_b=r(2 * inner, sc=0.1),
             fc2_w=r(C, inner, sc=inner ** -0.5), fc2_b=r(C, sc=0.1), ls2=0.5 + r(C, sc=0.1))
    return p


def timed(fn, reps):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


p = params()
x = torch.randn(B, H, W, C, generator=g).to(dev)
wgt = torch.randn(B, H, W, C, generator=g).to(dev)
for mode in ((0,) if args.quick else (0, 1)):
    for sharp in ((0.0, 0.3) if args.quick else (0.0, 0.3, 0.8)):      # token-score contrast: 0 = every token kept, larger = sparser
        tok = (torch.randn(B, H * W, generator=g) * sharp).exp().to(dev).contiguous()
        sel = SF.select(tok, B, H, W, ph, pw, mode, 0.0)
        kept = sel.num_kept_tokens()
        res = {}
        for fused in (True, False):
            SF._FUSED_ENABLE = fused
            torch.cuda.empty_cache()
            junk = torch.full((512 << 20,), float('nan'), device=dev); del junk      # scratch handed out next is NaN, not the other form's leftovers
            if args.bwd:
                xs = x.clone().requires_grad_(True)
                ps = {k: v.clone().requires_grad_(True) for k, v in p.items()}
                out = SF.mswsa(xs, sel, 1e-5, ps)
                (out * wgt).sum().backward()
                res[fused] = (out.detach(), xs.grad, {k: v.grad for k, v in ps.items()})

                def step():
                    o = SF.mswsa(xs, sel, 1e-5, ps)
                    (o * wgt).sum().backward()
                us = 0.0 if args.quick else timed(step, args.reps)
            else:
                with torch.no_grad():
                    out = SF.mswsa(x, sel, 1e-5, p)
                    res[fused] = (out, None, {})
                    us = timed(lambda: SF.mswsa(x, sel, 1e-5, p), args.reps)
            res[fused] += (us,)
        o0, o1 = res[False][0], res[True][0]
        line = f"mode {mode} sharp {sharp}: kept {kept}/{B * H * W}  out max|d| {float((o0 - o1).abs().max()):.3e} (|out| {float(o0.abs().max()):.2f})  unfused {res[False][3]:.1f} us  fused {res[True][3]:.1f} us"
        if args.bwd:
            worst = max(((float((res[False][2][k] - res[True][2][k]).abs().max()) / (float(res[False][2][k].abs().max()) + 1e-12)), k) for k in p)
            dx = float((res[False][1] - res[True][1]).abs().max()) / float(res[False][1].abs().max())
            line += f"  dx rel {dx:.2e}  worst param grad rel {worst[0]:.2e} ({worst[1]})"
        print(line, flush=True)
