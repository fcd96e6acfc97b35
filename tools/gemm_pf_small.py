"""register-prefetch depth on the SMALL grids of the step (stage 3 / 4 linears, PAFPN level-32 convs as plain GEMMs, the sparse-regime
row counts): micro-benchmark entry sast_test_gemm_nt (csrc/k_test.hip), HIP-event time per launch, cold-ish operands (a different
weight buffer per repetition so the weights come from HBM / Infinity Cache as in the step, not from L2)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sast_amd import _lib as L
lib = L.tools_lib()
nt = lib.sast_test_gemm_nt; nt.restype = C.c_int; nt.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p]
dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
TILES = {0: "Small 64x64 PF2", 13: "SmallK2 PF2", 55: "SmallK2 PF4", 56: "SmallK2 PF6", 57: "Small PF4", 14: "SmallK4 PF2", 58: "SmallK4 PF4", 19: "ThinK4 PF2", 50: "ThinK4 PF4",
         51: "ThinK4 PF6", 52: "ThinK4 PF8", 18: "TinyK8 PF2", 53: "TinyK8 PF4", 54: "TinyK8 PF6"}
NW = 24                                            # rotating weight buffers (> L2 of one XCD for the larger shapes)


def run(M, N, K):
    a = torch.randn(M, K, device=dev); ws = [torch.randn(N, K, device=dev) for _ in range(NW)]; b = torch.zeros(N, device=dev)
    c = torch.empty(M, N, device=dev)
    ref = (a.double() @ ws[0].double().t()).float()
    res = []
    for t, name in TILES.items():
        rc = nt(a.data_ptr(), ws[0].data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, t, st); torch.cuda.synchronize()
        if rc:
            res.append((1e9, name, rc, 0)); continue
        err = float((c - ref).abs().max() / ref.abs().max())
        for i in range(NW): nt(a.data_ptr(), ws[i].data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, t, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for rep in range(3):
            for i in range(NW): nt(a.data_ptr(), ws[i].data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, t, st)
        e1.record(); torch.cuda.synchronize()
        res.append((e0.elapsed_time(e1) * 1e3 / (3 * NW), name, rc, err))
    best = min(r[0] for r in res)
    print(f"NT {M}x{N}x{K}: " + "  ".join(f"{n} {us:5.1f}{'*' if us == best else ''}" + (f" ERR {err:.0e}" if err > 5e-6 else "") for us, n, rc, err in res if rc == 0), flush=True)


for shape in [(960, 512, 1344), (960, 512, 2688), (960, 1536, 512), (960, 256, 2304), (960, 512, 512), (3840, 256, 672), (3840, 256, 256),
              (3840, 768, 256), (3840, 128, 1152), (215, 512, 1344), (570, 512, 1344), (570, 1536, 512), (2500, 256, 672), (15360, 128, 320)]:
    run(*shape)
