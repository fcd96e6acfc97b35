"""per-workgroup phase timeline of the paired (dW || dX) launch of one conv + BN + SiLU backward, inside the PRODUCT kernel.
Needs the timeline variant of the library:

    python -m sast_amd.build --out ab/libsast_hip_tl.so --flags -DSAST_TL_ENABLE
    SAST_LIB_PATH=ab/libsast_hip_tl.so python tools/conv_timeline.py [B H W Cin Cout k]

Stamps (100 MHz wall clock): 0 entry, 1 first tile in LDS, 2 k-loop done, 3 k-group fold done, 4 epilogue done; slot 6 = job
(1 = split-R weight gradient, 2 = activation gradient), slot 7 = hardware CU id."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sast_amd import _lib as L, functional as SF
lib = L.lib()
rd = lib.sast_tl_read; rd.restype = C.c_int; rd.argtypes = [C.c_void_p, C.c_int]
rs = lib.sast_tl_reset; rs.restype = C.c_int; rs.argtypes = []
B, H, W, Cin, Cout, k = [int(v) for v in sys.argv[1:7]] if len(sys.argv) >= 7 else (4, 24, 40, 128, 128, 3)
dev = torch.device("cuda:0")
torch.manual_seed(0)
w = (torch.randn(Cout, Cin, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last).requires_grad_(True)
g, b = torch.ones(Cout, device=dev, requires_grad=True), torch.zeros(Cout, device=dev, requires_grad=True)
rm, rv = torch.zeros(Cout, device=dev), torch.ones(Cout, device=dev)
pre_w = (torch.randn(Cin, Cin, 1, 1, device=dev) * 0.05).contiguous(memory_format=torch.channels_last).requires_grad_(True)
pg, pb = torch.ones(Cin, device=dev, requires_grad=True), torch.zeros(Cin, device=dev, requires_grad=True)
for rep in range(3):
    x = torch.randn(B, H, W, Cin, device=dev, requires_grad=True)       # a leaf: the conv's paired launch is the only GEMM of the backward
    y = SF.conv_bn_silu(x, w, g, b, rm, rv, k, 1, True)
    dy = torch.randn_like(y)
    torch.cuda.synchronize()
    rs()
    y.backward(dy)
    torch.cuda.synchronize()
buf = np.zeros((8192, 8), dtype=np.uint64)
assert rd(buf.ctypes.data, 8192) == 0
t = buf[:, :5].astype(np.int64); job = buf[:, 6].astype(np.int64); cu = buf[:, 7].astype(np.int64)
ok = t[:, 4] > 0
t0 = t[ok][:, 0].min()
print(f"conv {k}x{k} {Cin}->{Cout} on {B}x{H}x{W}: {ok.sum()} workgroups ran; kernel span {(t[ok][:, 4].max() - t0) / 100.0:.1f} us")
for j, name in ((1, "dW split-R job"), (2, "dX job")):
    m = ok & (job == j)
    if not m.any():
        continue
    us = (t[m] - t0) / 100.0
    d = np.diff(us, axis=1)
    print(f"  {name}: {m.sum()} blocks | start p50 {np.median(us[:, 0]):.1f} max {us[:, 0].max():.1f} | end p50 {np.median(us[:, 4]):.1f} max {us[:, 4].max():.1f} us | "
          f"phases p50 prologue {np.median(d[:, 0]):.1f} loop {np.median(d[:, 1]):.1f} fold {np.median(d[:, 2]):.1f} epilogue {np.median(d[:, 3]):.1f} | "
          f"p90 {np.percentile(d[:, 0], 90):.1f} {np.percentile(d[:, 1], 90):.1f} {np.percentile(d[:, 2], 90):.1f} {np.percentile(d[:, 3], 90):.1f}")
# co-residency: how many blocks of each job shared a CU
ids = cu[ok]
from collections import Counter
cnt = Counter(ids.tolist())
print(f"  CUs used {len(cnt)}; blocks per CU histogram {sorted(Counter(cnt.values()).items())}")
