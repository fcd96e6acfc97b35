"""large tiles + a GRID-level split of the reduction (stream-K in its simplest form: equal k-ranges, atomic epilogue into a zeroed
output) against the shipped tiles, on the K-heavy mid-size shapes of stages 3 / 4 and the PAFPN.  micro-benchmark entry point
sast_test_gemm_nt (csrc/k_test.hip), HIP-event time per launch (the zero fill of the output is NOT included: + one 1-4 MB clear)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sast_amd import _lib as L
lib = L.tools_lib()
nt = lib.sast_test_gemm_nt; nt.restype = C.c_int; nt.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p]
dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
KIND = {6: "Big 128x128", 7: "Mid 64x128", 8: "N64 128x64"}
BASE = {0: "Small", 13: "SmallK2", 19: "ThinK4", 18: "TinyK8"}


def time(a, w, b, c, M, N, K, t):
    for _ in range(3): rc = nt(a.data_ptr(), w.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, t, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): nt(a.data_ptr(), w.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, t, st)
    e1.record(); torch.cuda.synchronize()
    return rc, e0.elapsed_time(e1) * 1e3 / 20


def run(M, N, K):
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.zeros(N, device=dev)
    ref = (a.double() @ w.double().t()).float()
    for t in (0, 13, 19, 18):
        c = torch.zeros(M, N, device=dev)
        rc, us = time(a, w, b, c, M, N, K, t)
        print(f"NT {M}x{N}x{K}  {BASE[t]:22s} rc {rc} {us:7.1f} us {2*M*N*K/us/1e6:6.1f} TF/s", flush=True)
    for kind, (bm, bn) in ((6, (128, 128)), (7, (64, 128)), (8, (128, 64))):
        tiles = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
        for splits in (1, 2, 4, 8, 16):
            if splits > 1 and (tiles * splits > 1100 or K // splits < 64):
                continue
            c = torch.zeros(M, N, device=dev)
            nt(a.data_ptr(), w.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, 100 * kind + splits, st); torch.cuda.synchronize()
            err = float((c - ref).abs().max() / ref.abs().max())
            rc, us = time(a, w, b, c, M, N, K, 100 * kind + splits)
            print(f"NT {M}x{N}x{K}  {KIND[kind]:12s} x{splits:2d} ({tiles * splits:4d} wg) rc {rc} {us:7.1f} us {2*M*N*K/us/1e6:6.1f} TF/s  err {err:.1e}", flush=True)


for shape in [(3840, 256, 1344), (3840, 256, 672), (3840, 768, 256), (960, 512, 2688), (960, 512, 1344), (960, 1536, 512), (3840, 128, 1152), (3840, 256, 2304),
              (960, 2048, 512), (3840, 1024, 256)]:
    run(*shape)
