import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import torch
from sast_amd import _lib as L
lib = L.tools_lib(); fn = lib.sast_test_gemm_nt; fn.restype = C.c_int
fn.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p]
dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
shapes = [(960, 256, 2304), (960, 128, 1152), (960, 256, 512), (3840, 128, 1152), (3840, 64, 576), (15360, 64, 576), (15360, 128, 1152), (960, 1536, 512), (960, 512, 512), (960, 2688, 512), (960, 512, 1344), (960, 512, 1536), (960, 512, 2688), (3840, 768, 256), (3840, 256, 768), (3840, 256, 1344), (3840, 1344, 256), (960, 2048, 1024)]
tiles = [(0, "64x64"), (13, "K2"), (14, "K4"), (19, "32x64K4"), (17, "32x32K4"), (18, "32x32K8")]
print("M,N,K".ljust(16) + " ".join(n.rjust(14) for _, n in tiles))
for (M, N, K) in shapes:
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev); c = torch.empty(M, N, device=dev)
    line = f"{M}x{N}x{K}".ljust(16)
    for t, _ in tiles:
        for _ in range(3): fn(a.data_ptr(), w.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, t, st)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            sp = torch.cuda.current_stream().cuda_stream
            with torch.cuda.graph(g, stream=s):
                for _ in range(20): fn(a.data_ptr(), w.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, t, sp)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g.replay(); torch.cuda.synchronize()
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        line += f" {us:7.1f}us{2.0 * M * N * K / us / 1e6:5.0f}"
    print(line)
