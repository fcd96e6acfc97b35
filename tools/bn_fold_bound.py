"""Upper bound of what folding BatchNorm-apply + SiLU into the A-loaders of the 1x1 consumers could buy (round-2 verdict item 7),
measured BEFORE building it: a timing-only variant of the library (-DSAST_EXP_SKIP_FOLDABLE_BN_APPLY, results are wrong) drops the
forward apply launch of every PAFPN conv whose output is consumed by 1x1 convs only -- exactly the launches (and the y write + re-read)
the fold would remove, with none of the costs it would add (sigmoid + table lookups in the consumers' forward and weight-gradient
loaders).  Foldable in the shipped PAFPN (depth 0.67, n = 2 bottlenecks per CSP layer, yolo_pafpn.py / network_blocks.py):
per CSP layer conv1 + conv2 (one shared pass), m[0].conv2, m[1].conv2; plus C3_p4.conv3 (-> reduce_conv1) and bu_conv2 (-> C3_n3 conv1/2).

    python -m sast_amd.build --out ab/bnskip.so --flags "-DSAST_EXP_SKIP_FOLDABLE_BN_APPLY"
    SAST_LIB_PATH=ab/bnskip.so python tools/bn_fold_bound.py
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sast_amd import functional as SF


def timed(tr, steps=100, warm=20):
    for _ in range(warm): tr.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): tr.step()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


def main():
    dev = torch.device("cuda:0")
    s = torch.cuda.Stream(); torch.cuda.set_stream(s)
    res = {}
    for mode in ("all applies", "foldable applies dropped", "all applies", "foldable applies dropped"):
        tr = bench.Trainer(dev, 2e-4, 1, use_graph=True)
        fpn = tr.fpn
        fold = set()
        for csp in (fpn.C3_p4, fpn.C3_p3, fpn.C3_n3, fpn.C3_n4):
            fold |= {id(csp.conv1.conv.weight), id(csp.conv2.conv.weight)} | {id(b.conv2.conv.weight) for b in csp.m}
        fold |= {id(fpn.C3_p4.conv3.conv.weight), id(fpn.bu_conv2.conv.weight)}
        drop = mode.startswith("foldable")
        o1, o2 = SF.conv_bn_silu, SF.conv_bn_silu2
        n = [0]

        def p1(x, w, bn_w, bn_b, rm, rv, k, st, training, momentum=0.1, eps=1e-5, bn_ws=None, **kw):
            if drop and id(w) in fold:
                momentum = -abs(momentum); n[0] += 1
            return o1(x, w, bn_w, bn_b, rm, rv, k, st, training, momentum, eps, bn_ws, **kw)

        def p2(x, c0, c1, bn_ws=(None, None), **kw):
            if drop and id(c0[0]) in fold:
                c0 = c0[:5] + (-abs(c0[5]),) + c0[6:]; n[0] += 1
            return o2(x, c0, c1, bn_ws, **kw)

        SF.conv_bn_silu, SF.conv_bn_silu2 = p1, p2
        try:
            tr.capture()
            ms = timed(tr)
        finally:
            SF.conv_bn_silu, SF.conv_bn_silu2 = o1, o2
        print(f"{mode:28s} {ms:7.4f} ms/step   (apply launches dropped per forward: {n[0] // 3 if drop else 0})", flush=True)
        res.setdefault(mode, []).append(ms)
        del tr
        torch.cuda.empty_cache()
    a, b = min(res["all applies"]), min(res["foldable applies dropped"])
    print(f"upper bound of the fold: {a - b:.4f} ms/step ({100 * (a - b) / a:.2f} %)")


if __name__ == "__main__":
    main()
