"""A weight-stationary backward-data GEMM (csrc/k_ws_test.hip: c = a w, the weight split ONCE per workgroup into LDS-resident bf16x3 planes,
activation rows streamed from global memory straight into MFMA operand registers; no barrier / LDS store / weight split in the k-loop)
against the shipped tiles of the GEMM template on the shapes of the dim-64 / dim-128 layers.  Time per launch inside a replayed hipGraph of
20 launches; errors against fp64."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sast_amd import _lib as L
lib = L.tools_lib()
ws = lib.sast_test_ws_gemm_nn; ws.restype = C.c_int; ws.argtypes = [C.c_void_p] * 3 + [C.c_int] * 5 + [C.c_void_p]
nn = lib.sast_test_gemm_nn; nn.restype = C.c_int; nn.argtypes = [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_void_p]
dev = torch.device("cuda:0")
SHIPPED = {0: "64x64", 1: "64x64 k2", 2: "32x64 k4", 3: "128x64", 9: "gemm_auto"}


def timeit(fn, reps=20):
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        rc = fn(st)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cs = torch.cuda.current_stream().cuda_stream
        for _ in range(reps):
            fn(cs)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return rc, e0.elapsed_time(e1) * 1e3 / (5 * reps)


def run(M, N, K):
    a = torch.randn(M, K, device=dev); w = torch.randn(K, N, device=dev)
    ref = a.double() @ w.double()
    scale = float(ref.abs().max())
    best = None
    for t, name in SHIPPED.items():
        c = torch.zeros(M, N, device=dev)
        rc, us = timeit(lambda s_: nn(a.data_ptr(), w.data_ptr(), c.data_ptr(), M, N, K, t, s_))
        err = float((c.double() - ref).abs().max()) / scale
        if rc == 0:
            best = us if best is None else min(best, us)
        print(f"NN {M}x{N}x{K}  shipped {name:9s} rc {rc} {us:7.1f} us {2*M*N*K/us/1e6:6.1f} TF/s  err {err:.1e}", flush=True)
    for waves in (4, 8):
        for blocks in (256, 512, 768, 1024):
            c = torch.zeros(M, N, device=dev)
            rc = ws(a.data_ptr(), w.data_ptr(), c.data_ptr(), M, N, K, waves, blocks, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            if rc != 0:
                print(f"NN {M}x{N}x{K}  weight-stationary waves {waves} blocks {blocks} rc {rc}", flush=True)
                continue
            err = float((c.double() - ref).abs().max()) / scale
            _rc, us = timeit(lambda s_: ws(a.data_ptr(), w.data_ptr(), c.data_ptr(), M, N, K, waves, blocks, s_))
            print(f"NN {M}x{N}x{K}  weight-stationary {waves} waves x {blocks:4d} blocks {us:7.1f} us {2*M*N*K/us/1e6:6.1f} TF/s  err {err:.1e}  "
                  f"({us / best:.2f} x the best shipped tile)", flush=True)


if __name__ == "__main__":
    shapes = [(61440, 64, 192), (61440, 64, 64), (61440, 64, 320), (15360, 128, 128), (15360, 128, 384), (20480, 64, 192), (122880, 64, 192)]
    if len(sys.argv) > 3:
        shapes = [tuple(int(v) for v in sys.argv[1:4])]
    for s in shapes:
        run(*s)
