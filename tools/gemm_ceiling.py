import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import torch
from sast_amd import _lib as L
lib = L.tools_lib(); fn = lib.sast_test_gemm_nt; fn.restype = C.c_int
fn.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p]
dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
for (M, N, K) in [(4096, 4096, 4096), (8192, 8192, 2048), (16384, 2048, 2048)]:
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev); c = torch.empty(M, N, device=dev)
    line = f"{M}x{N}x{K}".ljust(18)
    for t, nm in [(0, "64x64"), (1, "64x128"), (2, "128x128"), (13, "64x64K2"), (8, "128x64")]:
        for _ in range(2): fn(a.data_ptr(), w.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, t, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn(a.data_ptr(), w.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, t, st)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 5
        line += f" {nm}: {2.0 * M * N * K / us / 1e6:5.0f}TF"
    # torch (rocBLAS/hipBLASLt) fp32 reference point
    for _ in range(2): torch.mm(a, w.t())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): torch.mm(a, w.t())
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 5
    line += f" | torch.mm fp32: {2.0 * M * N * K / us / 1e6:5.0f}TF"
    print(line)
