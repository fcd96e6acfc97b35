"""what the GEMM template reaches on LARGE problems and with LARGE tiles under the exact bf16x3 operand split (6 MFMAs per product):
the ceiling of the engine, next to the mid-size shapes of stages 3 / 4 and the PAFPN on the same tiles.  micro-benchmark entry
point sast_test_gemm_nt (csrc/k_test.hip); HIP-event time per launch."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sast_amd import _lib as L
lib = L.tools_lib()
nt = lib.sast_test_gemm_nt; nt.restype = C.c_int; nt.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p]
dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
NAMES = {0: "Small 64x64 (4w 32x32)", 1: "Mid 64x128 (4w 32x64)", 2: "Big 128x128 (4w 64x64)", 3: "N64 128x64 (4w 32x64)", 8: "128x64 2x2 (64x32)", 13: "SmallK2", 19: "ThinK4"}


def run(M, N, K, tiles):
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    for t in tiles:
        c = torch.zeros(M, N, device=dev)
        for _ in range(3): rc = nt(a.data_ptr(), w.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, t, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): nt(a.data_ptr(), w.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, t, st)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print(f"NT {M}x{N}x{K} tile {t:2d} {NAMES.get(t, ''):24s} rc {rc} {us:8.1f} us  {2*M*N*K/us/1e6:6.1f} TF/s", flush=True)


run(4096, 4096, 4096, [0, 1, 2])
run(8192, 8192, 2048, [0, 1, 2])
run(16384, 1024, 1024, [0, 1, 2])
for shape in [(3840, 768, 256), (3840, 2688, 256), (3840, 256, 672), (3840, 256, 1344), (960, 1536, 512), (960, 5376, 512), (960, 512, 1344), (960, 512, 2688),
              (3840, 128, 1152), (15360, 640, 128), (15360, 128, 320)]:
    run(*shape, [0, 13, 19, 1, 2, 3])
