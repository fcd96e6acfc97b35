"""Per-phase shader-clock timeline of the fused MS-WSA forward kernel (a -DSAST_FUSED_TL build of the library: lane 0 of every wave
stamps s_memtime at the phase boundaries).  Prints the mean cycles per phase over the waves of one dense stage-1 launch.

    python -m sast_amd.build --out ab/fused_tl.so --flags "-DSAST_FUSED_TL"
    SAST_LIB_PATH=ab/fused_tl.so python tools/fused_timeline.py
"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sast_amd import functional as SF, _lib as L

dev = torch.device("cuda:0")
B, H, W, Cc, ph, pw, inner = 4, 96, 160, 64, 6, 10, 160
g = torch.Generator().manual_seed(1)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
p = dict(ln1_w=1 + r(Cc, sc=0.1), ln1_b=r(Cc, sc=0.1), ln2_w=1 + r(Cc, sc=0.1), ln2_b=r(Cc, sc=0.1), qkv_w=r(3 * Cc, Cc, sc=Cc ** -0.5), qkv_b=r(3 * Cc, sc=0.1),
         proj_w=r(Cc, Cc, sc=Cc ** -0.5), proj_b=r(Cc, sc=0.1), ls1=0.5 + r(Cc, sc=0.1), fc1_w=r(2 * inner, Cc, sc=Cc ** -0.5), fc1_b=r(2 * inner, sc=0.1),
         fc2_w=r(Cc, inner, sc=inner ** -0.5), fc2_b=r(Cc, sc=0.1), ls2=0.5 + r(Cc, sc=0.1))
x = r(B, H, W, Cc)
tok = torch.ones(B, H * W, device=dev)
sel = SF.select(tok, B, H, W, ph, pw, 0, 0.0)
with torch.no_grad():
    for _ in range(3):
        SF.mswsa(x, sel, 1e-5, p)
torch.cuda.synchronize()
nw = 2 * B * (H // ph) * (W // pw)      # one wave per tile of 32 kept tokens, two per partition
buf = (C.c_ulonglong * (24 * nw))()
fn = L.lib().sast_fused_tl_read
fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_int]
assert fn(buf, nw) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(nw, 24).astype(np.int64)
t = t[t[:, 20] > 0]
nw = len(t)
names = {0: "start", 1: "load+LN1+LN2", 2: "S operands", 3: "h0 QKV", 4: "h0 q/k/v operands", 5: "h0 attention", 6: "h0 proj", 7: "h1 QKV", 8: "h1 operands",
         9: "h1 attention", 10: "h1 proj", 11: "(heads done)", 12: "Y + Y operands", 13: "chunk 0", 14: "chunk 1", 15: "chunk 2", 16: "chunk 3", 17: "chunk 4",
         19: "(mlp done)", 20: "epilogue + store"}
order = [k for k in sorted(names)]
wall = (t[:, 22] - t[:, 21]) * 10.0      # s_memrealtime: 100 MHz
print(f"wall clock per wave: mean {np.mean(wall):.0f} ns; launch span {(np.max(t[:, 22]) - np.min(t[:, 21])) * 10.0:.0f} ns; "
      f"clock64 units per ns: {np.mean(t[:, 20] - t[:, 0]) / np.mean(wall):.3f}")
print(f"waves {nw}; total cycles per wave mean {np.mean(t[:, 20] - t[:, 0]):.0f} (min {np.min(t[:, 20] - t[:, 0])}, max {np.max(t[:, 20] - t[:, 0])}); "
      f"launch span {np.max(t[:, 20]) - np.min(t[:, 0])} cycles")
prev = 0
for k in order[1:]:
    d = t[:, k] - t[:, prev]
    print(f"  {names[k]:22s} {np.mean(d):9.0f} cycles  (min {np.min(d)}, max {np.max(d)})")
    prev = k
