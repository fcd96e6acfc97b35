"""Bounding experiment for the round-4 verdict's item 4: how fast is a mid-size forward GEMM (C = A W^T + b) when its k-loop has no VGPR
staging, no LDS stores, no barrier -- the weight operand pre-split into bf16x3 planes in MFMA operand order (a "weight image" in HBM),
the activation operand plain fp32 rows, both moved global -> LDS by global_load_lds_dwordx4 into wave-private rings (csrc/k_dma_test.hip)
-- against the SHIPPED tiles of the GEMM template on the same shapes (csrc/k_test.hip: sast_test_gemm_nt)?

Time per launch inside a replayed hipGraph of 20 launches (L2-warm; --eager: event-bracketed eager launches); errors against fp64.  cfg = 100 TN + 10 log2(ring depth) + log2(k-groups)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sast_amd import _lib as L
lib = L.tools_lib()
nt = lib.sast_test_gemm_nt; nt.restype = C.c_int; nt.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p]
img_bytes = lib.sast_test_weight_image_bytes; img_bytes.restype = C.c_size_t; img_bytes.argtypes = [C.c_int, C.c_int]
mk_img = lib.sast_test_weight_image; mk_img.restype = C.c_int; mk_img.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
dma = lib.sast_test_dma_gemm_nt; dma.restype = C.c_int; dma.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p]
dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
SHIPPED = {13: "64x64 k2", 19: "32x64 k4", 18: "32x32 k8", 0: "64x64", 1: "64x128"}
CFGS = [122, 121, 222, 212, 221, 220, 223, 422]
DIAG = {1212: "k order rotated per workgroup (ring 2)", 1222: "k order rotated per workgroup (ring 4)", 2212: "no A pieces", 4212: "no weight pieces",
        8212: "no MFMA / split (data movement only)", 10212: "weights only, no MFMA", 12212: "A only, no MFMA", 6212: "no loads at all (LDS reads + MFMA)"}


GRAPH = "--eager" not in sys.argv


def timeit(fn, reps=20):
    """fn(stream) launches once.  Default: `reps` launches captured into ONE hipGraph, replayed 5 times -- the time per launch inside a
    replayed graph is what the product's step pays (kernel-node floor 1.6 us); --eager: back-to-back eager launches between two events,
    which for kernels under ~10 us measures the host's launch rate (ctypes + hipLaunchKernel), not the kernel."""
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        rc = fn(st)
    torch.cuda.synchronize()
    if not GRAPH:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn(st)
        e1.record(); torch.cuda.synchronize()
        return rc, e0.elapsed_time(e1) * 1e3 / reps
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
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    ref = a.double() @ w.double().t() + b.double()
    scale = float(ref.abs().max())
    best = None
    for t, name in SHIPPED.items():
        c = torch.zeros(M, N, device=dev)
        rc, us = timeit(lambda s_: nt(a.data_ptr(), w.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, t, s_))
        err = float((c.double() - ref).abs().max()) / scale
        best = us if best is None else min(best, us)
        print(f"NT {M}x{N}x{K}  shipped {name:9s} rc {rc} {us:7.1f} us {2*M*N*K/us/1e6:6.1f} TF/s  err {err:.1e}", flush=True)
    img = torch.empty(img_bytes(N, K), device=dev, dtype=torch.uint8)
    assert mk_img(w.data_ptr(), N, K, img.data_ptr(), st) == 0
    _rc, us_img = timeit(lambda s_: mk_img(w.data_ptr(), N, K, img.data_ptr(), s_))
    print(f"NT {M}x{N}x{K}  weight image build {us_img:6.1f} us ({img.numel() / 1e6:.2f} MB)", flush=True)
    for cfg in CFGS:
        tn = cfg // 100
        if N % (32 * tn):
            continue
        c = torch.zeros(M, N, device=dev)
        rc = dma(a.data_ptr(), img.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, cfg, st)
        torch.cuda.synchronize()
        if rc != 0:
            print(f"NT {M}x{N}x{K}  dma cfg {cfg} rc {rc}", flush=True)
            continue
        err = float((c.double() - ref).abs().max()) / scale
        rc, us = timeit(lambda s_: dma(a.data_ptr(), img.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, cfg, s_))
        waves = ((M + 31) // 32) * (N // (32 * tn)) * (1 << (cfg % 10))
        print(f"NT {M}x{N}x{K}  dma TN{tn} ring{1 << ((cfg // 10) % 10)} kg{1 << (cfg % 10)} ({waves:5d} waves) rc {rc} {us:7.1f} us {2*M*N*K/us/1e6:6.1f} TF/s  err {err:.1e}  "
              f"vs best shipped {us / best:5.2f}x", flush=True)


    for cfg, what in DIAG.items():
        c = torch.zeros(M, N, device=dev)
        rc = dma(a.data_ptr(), img.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, cfg, st)
        torch.cuda.synchronize()
        if rc != 0:
            print(f"NT {M}x{N}x{K}  diag {cfg} rc {rc}", flush=True)
            continue
        err = float((c.double() - ref).abs().max()) / scale
        rc, us = timeit(lambda s_: dma(a.data_ptr(), img.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, cfg, s_))
        print(f"NT {M}x{N}x{K}  diag TN2 kg4: {what:42s} {us:7.1f} us  (err {err:.1e}{'' if cfg < 2000 else ': wrong by design'})", flush=True)


shapes = [(3840, 128, 1152), (960, 1344, 512), (960, 512, 1344), (3840, 256, 672), (3840, 768, 256), (3840, 256, 256), (960, 512, 512),
          (960, 1536, 512), (960, 256, 2304), (15360, 384, 128), (15360, 128, 128)]
args = [v for v in sys.argv[1:] if not v.startswith("--")]
if args:
    shapes = [tuple(int(v) for v in s.split("x")) for s in args]
for shape in shapes:
    run(*shape)
