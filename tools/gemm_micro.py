"""GEMM-template micro-benchmark on the shapes of the SAST workload (1Mpx, B=4)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sast_amd import _lib as L
lib = L.tools_lib()
fn = lib.sast_test_gemm_nt
fn.restype = C.c_int
fn.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p]
dev = torch.device("cuda:0")
shapes = [(61440, 192, 64), (61440, 64, 64), (61440, 320, 64), (61440, 64, 160), (15360, 384, 128), (15360, 128, 128), (15360, 640, 128),
          (15360, 128, 320), (3840, 768, 256), (3840, 256, 256), (3840, 1344, 256), (3840, 256, 672), (960, 1536, 512), (960, 512, 512),
          (960, 2688, 512), (960, 512, 1344), (61440, 256, 128), (15360, 512, 256), (3840, 1024, 512), (960, 2048, 1024), (3840, 128, 1152), (15360, 64, 576), (960, 256, 2304), (15360, 128, 1152), (3840, 256, 2304)]
tiles = {0: "64x64k16", 13: "64x64 KS2", 19: "32x64 KS4"}
st = torch.cuda.current_stream().cuda_stream
print("shape".ljust(22) + " ".join(f"{v:>15}" for v in tiles.values()))
for (M, N, K) in shapes:
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev); c = torch.empty(M, N, device=dev)
    ref = a @ w.t() + b
    line = f"{M}x{N}x{K}".ljust(22)
    for t in tiles:
        rc = fn(a.data_ptr(), w.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, t, st)
        assert rc == 0
        err = float((c - ref).abs().max())
        for _ in range(3):
            fn(a.data_ptr(), w.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, t, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn(a.data_ptr(), w.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, t, st)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        tf = 2.0 * M * N * K / us / 1e6
        line += f" {us:7.1f}us{tf:5.0f}TF" + ("!" if err > 1e-2 else " ")
    print(line)
