"""Round 6: what do the saved-activation stores cost the one-kernel MS-WSA forward?  Times sast_mswsa_fwd (fused form) on the 1Mpx stage-1
shape in its inference form (no saves) and its training form (S / QKV / O / Y / [u|g] / h / lse written), replayed from a hipGraph.
Run with SAST_LIB_PATH=ab/nowait.so for the timing-only build whose weight ring does not wait (wrong results)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sast_amd import functional as SF
from sast_amd.layers.sast import SAST_block  # noqa: F401  (module import registers nothing; kept for parity with the other tools)

dev = torch.device("cuda:0")
B, H, W, C, ph, pw, inner = 4, 96, 160, 64, 6, 10, 160
torch.manual_seed(0)
x = torch.randn(B, H, W, C, device=dev)
tok = torch.ones(B, H * W, device=dev)
sel = SF.select(tok, B, H, W, ph, pw, 0, 1e-3)          # uniform scores: every window and token is kept (the dense headline regime)
p = {"ln1_w": torch.ones(C), "ln1_b": torch.zeros(C), "ln2_w": torch.ones(C), "ln2_b": torch.zeros(C), "qkv_w": torch.randn(3 * C, C) * 0.05,
     "qkv_b": torch.zeros(3 * C), "proj_w": torch.randn(C, C) * 0.05, "proj_b": torch.zeros(C), "ls1": torch.full((C,), 0.5),
     "fc1_w": torch.randn(2 * inner, C) * 0.05, "fc1_b": torch.zeros(2 * inner), "fc2_w": torch.randn(C, inner) * 0.05, "fc2_b": torch.zeros(C),
     "ls2": torch.full((C,), 0.5)}
p = {k: v.to(dev).requires_grad_(True) for k, v in p.items()}
SF._FUSED_MIN_ROWS = 0


def timed(train):
    def f():
        if train:
            return SF.mswsa(x, sel, 1e-5, p)
        with torch.no_grad():
            return SF.mswsa(x, sel, 1e-5, p)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(10):
                f()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / 200


print(os.environ.get("SAST_LIB_PATH", "product"), "kept rows", int(sel.num_kept_tokens()), "of", B * H * W,
      "| forward per call (planes + layer kernel): inference form %.1f us, training form %.1f us" % (timed(False), timed(True)))
