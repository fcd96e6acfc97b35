"""tile shapes of the GEMM template against each other on the model's shapes: y[M,N] = x[M,K] W[N,K]^T + b through the micro-benchmark entry
point (csrc/k_test.hip), checked against fp64, HIP-event time per launch.  SAST_TOOLS_LIB_PATH selects a variant build of the tools library
(-DSAST_MFMA_BF16=1, -DSAST_MFMA_SPLIT3=0, -DSAST_PRESPLIT_RC=0 ...)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sast_amd import _lib as L
lib = L.tools_lib()
nt = lib.sast_test_gemm_nt; nt.restype = C.c_int; nt.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p]
dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
NAMES = {40: "Small BK32", 41: "SmallK2 BK32", 42: "ThinK2 BK32", 43: "ThinK4 BK32", 44: "Mid BK32", 0: "Small(4w 64x64)", 13: "SmallK2", 19: "ThinK4(2w x4)", 18: "TinyK8(1w x8)", 17: "32x32 K4", 30: "W64x64", 31: "W64x64 K2", 32: "W64x64 K4", 33: "W32x64", 34: "W32x64 K2", 35: "W32x64 K4", 36: "W32x32 K2", 1: "Mid", 9: "Tiny(1w 32x32)"}
def run(M, N, K, tiles):
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    ref = (a.double() @ w.double().t() + b.double()).float()
    for t in tiles:
        c = torch.zeros(M, N, device=dev)
        rc = nt(a.data_ptr(), w.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, t, st)
        torch.cuda.synchronize()
        err = float((c - ref).abs().max() / ref.abs().max())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5): nt(a.data_ptr(), w.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, t, st)
        e0.record()
        for _ in range(50): nt(a.data_ptr(), w.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, t, st)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 50
        print(f"NT {M}x{N}x{K} tile {t:2d} {NAMES.get(t, ''):18s} rc {rc} {us:7.1f} us  {2*M*N*K/us/1e6:6.1f} TF/s  err {err:.1e}")
NAMES.update({3: "N64 (128x64, 4w)", 2: "Big 128x128", 8: "128x64 2x2"})
run(61440, 192, 64, [0, 3, 8, 1])
run(61440, 64, 160, [0, 13, 3, 8])
run(61440, 64, 64, [0, 3, 8])
run(61440, 320, 64, [0, 3, 1, 2])
run(61440, 64, 320, [0, 13, 3, 8])
run(15360, 384, 128, [0, 3, 1, 2])
run(15360, 128, 128, [0, 3, 8])
run(15360, 128, 640, [13, 0, 3, 8])
