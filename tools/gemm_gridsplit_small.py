"""the SHIPPED small tiles (32x64 with 4 k-groups, 32x32 with 8, 64x64 with 2) with an ADDITIONAL grid-level split of the reduction
(equal k-ranges, atomic epilogue into a zeroed output) on the small-M shapes of stages 3 / 4 and at B = 1 -- the question the round-3
verdict's item 2 leaves open after `gemm_gridsplit.py` (large tiles): does more k-parallelism per output tile shorten the dependent chain of
a launch that fills the chip only once?  micro-benchmark entry point sast_test_gemm_nt (csrc/k_test.hip), HIP-event time per launch; the
zero fill of the output and the epilogue fix-up a real integration needs are NOT included (lower bound on the cost)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sast_amd import _lib as L
lib = L.tools_lib()
nt = lib.sast_test_gemm_nt; nt.restype = C.c_int; nt.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p]
dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
KIND = {9: ("32x64 k4", 32, 64, 4), 10: ("32x32 k8", 32, 32, 8), 11: ("64x64 k2", 64, 64, 2)}
BASE = {13: "SmallK2 64x64 k2", 19: "ThinK4 32x64 k4", 18: "TinyK8 32x32 k8"}


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
    best = None
    for t in (13, 19, 18):
        c = torch.zeros(M, N, device=dev)
        rc, us = time(a, w, b, c, M, N, K, t)
        best = us if best is None else min(best, us)
        print(f"NT {M}x{N}x{K}  shipped {BASE[t]:18s} rc {rc} {us:7.1f} us {2*M*N*K/us/1e6:6.1f} TF/s", flush=True)
    for kind, (name, bm, bn, kg) in KIND.items():
        tiles = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
        for splits in (2, 4, 8):
            if tiles * splits > 2100 or K // (splits * kg) < 16:
                continue
            c = torch.zeros(M, N, device=dev)
            nt(a.data_ptr(), w.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, 100 * kind + splits, st); torch.cuda.synchronize()
            err = float((c - ref).abs().max() / ref.abs().max())
            rc, us = time(a, w, b, c, M, N, K, 100 * kind + splits)
            print(f"NT {M}x{N}x{K}  {name:10s} grid x{splits} ({tiles * splits:4d} wg) rc {rc} {us:7.1f} us {2*M*N*K/us/1e6:6.1f} TF/s  err {err:.1e}  vs best shipped {us / best:5.2f}x", flush=True)


for shape in [(960, 512, 512), (960, 1536, 512), (960, 2688, 512), (960, 512, 1344), (3840, 256, 256), (3840, 768, 256), (3840, 1344, 256),
              (3840, 256, 672), (3840, 128, 256), (240, 512, 512), (240, 2688, 512), (240, 512, 1344), (960, 256, 256), (960, 256, 672)]:
    run(*shape)
