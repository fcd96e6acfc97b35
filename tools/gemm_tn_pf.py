"""weight-gradient GEMM: prefetch depth / occupancy sweep.  Kernel time = last block end - first block start from the
in-kernel timeline (k_test.hip), i.e. without launch gaps."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sast_amd import _lib as L
lib = L.tools_lib()
tn = lib.sast_test_gemm_tn; tn.restype = C.c_int; tn.argtypes = [C.c_void_p] * 4 + [C.c_int] * 6 + [C.c_void_p]
tl = lib.sast_test_timeline; tl.restype = C.c_int; tl.argtypes = [C.c_void_p, C.c_int]
tlr = lib.sast_test_timeline_reset; tlr.restype = C.c_int; tlr.argtypes = []
dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
VAR = {1: "K2 2x2w", 0: "K1 2x2w", 50: "K2 32x64w", 51: "K4 32x64w", 52: "K4 64x32w", 53: "K4 64x64w", 54: "K8 64x64w", 30: "128x64"}
TSZ_EXTRA = {33: (128, 128)}
TSZ = {30: (128, 64), 35: (128, 64), 36: (128, 64), 32: (64, 128), 33: (128, 128), 34: (192, 64)}
def span(nblocks):
    buf = np.zeros((nblocks, 8), dtype=np.uint64)
    assert tl(buf.ctypes.data, nblocks) == 0
    t = buf[:, :5].astype(np.int64)
    t = t[t[:, 4] > 0]
    return (t[:, 4].max() - t[:, 0].min()) / 100.0
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(192, 64, 61440), (64, 64, 61440), (320, 64, 61440), (64, 160, 61440), (384, 128, 15360), (640, 128, 15360), (128, 320, 15360), (768, 256, 3840), (1344, 256, 3840), (256, 672, 3840), (1536, 512, 960), (2688, 512, 960), (512, 1344, 960), (512, 512, 960)]
print("shape".ljust(18) + "blocks " + " ".join(v.rjust(10) for v in VAR.values()))
for (Mo, NJ, R) in shapes:
    dy = torch.randn(R, Mo, device=dev); x = torch.randn(R, NJ, device=dev); out = torch.zeros(Mo, NJ, device=dev); cs = torch.zeros(Mo, device=dev)
    ref = dy.t() @ x
    for total in (152, 256, 512, 768):
        line = f"{Mo}x{NJ}x{R}".ljust(18) + f"{total:5d}  "
        for t in VAR:
            bm, bn = TSZ.get(t, (64, 64))
            nb = ((Mo + bm - 1) // bm) * ((NJ + bn - 1) // bn)
            splits = max(1, min((total + nb - 1) // nb, (R + 127) // 128))
            best = 1e9
            rc = 0
            for rep in range(4):
                out.zero_(); cs.zero_(); tlr()
                rc = tn(dy.data_ptr(), x.data_ptr(), out.data_ptr(), cs.data_ptr(), Mo, NJ, R, t, splits, 0, st)
                if rc:
                    break
                torch.cuda.synchronize()
                best = min(best, span(min(8192, nb * splits + 8)))
            if rc:
                line += f"   rc={rc:4d}"
                continue
            if t < 100:
                err = float((out - ref).abs().max() / ref.abs().max())
                assert err < 1e-4, (t, err)
            line += f" {best:8.1f}us"
        print(line)
