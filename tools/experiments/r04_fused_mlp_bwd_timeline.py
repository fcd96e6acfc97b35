"""Per-phase shader-clock timeline of the fused MLP-backward kernel (-DSAST_FUSED_TL build): mean cycles per phase over the waves of one
dense stage-1 launch (61 440 rows).
    python -m sast_amd.build --out ab/fused_tl.so --flags "-DSAST_FUSED_TL"; SAST_LIB_PATH=$PWD/ab/fused_tl.so python tools/fused_mlp_bwd_timeline.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sast_amd import functional as SF, _lib as L
dev = torch.device("cuda:0")
B, H, W, Cc, ph, pw, inner = 4, 96, 160, 64, 6, 10, 160
g = torch.Generator().manual_seed(1)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev).requires_grad_(True)
p = dict(ln1_w=r(Cc), ln1_b=r(Cc, sc=0.1), ln2_w=r(Cc), ln2_b=r(Cc, sc=0.1), qkv_w=r(3 * Cc, Cc, sc=Cc ** -0.5), qkv_b=r(3 * Cc, sc=0.1),
         proj_w=r(Cc, Cc, sc=Cc ** -0.5), proj_b=r(Cc, sc=0.1), ls1=r(Cc, sc=0.5), fc1_w=r(2 * inner, Cc, sc=Cc ** -0.5), fc1_b=r(2 * inner, sc=0.1),
         fc2_w=r(Cc, inner, sc=inner ** -0.5), fc2_b=r(Cc, sc=0.1), ls2=r(Cc, sc=0.5))
x = r(B, H, W, Cc)
sel = SF.select(torch.ones(B, H * W, device=dev), B, H, W, ph, pw, 0, 0.0)
for _ in range(3):
    SF.mswsa(x, sel, 1e-5, p).sum().backward()
torch.cuda.synchronize()
nw = B * H * W // 32
buf = (C.c_ulonglong * (24 * nw))()
fn = L.lib().sast_fused_tl_read; fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_int]
assert fn(buf, nw) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(nw, 24).astype(np.int64)
t = t[t[:, 23] > 0]
print(f"waves {len(t)}; cycles per wave mean {np.mean(t[:, 23] - t[:, 0]):.0f}")
names = {1: "prologue (loads, operands, transposes of Y / dZ)"}
for kc in range(5):
    names[2 + 4 * kc] = f"chunk {kc}: fc1 recompute + dH MFMAs"; names[3 + 4 * kc] = f"chunk {kc}: gelu, dU dG, dY MFMAs"
    names[4 + 4 * kc] = f"chunk {kc}: transposes, dW MFMAs, partial tiles parked"; names[5 + 4 * kc] = f"chunk {kc}: barrier, owner sums + atomics, barrier"
names[22] = "dY store + barrier"; names[23] = "db1 / colsum(dZ) flush"
prev = 0
for k in sorted(names):
    d = t[:, k] - t[:, prev]
    print(f"  {names[k]:52s} {np.mean(d):9.0f} cycles (min {np.min(d)}, max {np.max(d)})")
    prev = k
