"""where the cycles of a k-loop phase go: shader-clock stamps in wave 0 of every block of the micro-benchmark GEMM (csrc/k_test.hip built with
-DSAST_TLF_ENABLE: `hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSAST_TLF_ENABLE -shared sast_amd/csrc/k_test.hip -o ab/libtools_tlf.so
-Lsast_amd -lsast_hip -Wl,-rpath,$PWD/sast_amd`).  The stamps serialise the schedule: read the shares, not the absolute cycles."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SAST_TOOLS_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ab", "libtools_tlf.so"))
import numpy as np, torch
from sast_amd import _lib as L
lib = L.tools_lib()
nt = lib.sast_test_gemm_nt; nt.restype = C.c_int; nt.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p]
tlf = lib.sast_test_tlf; tlf.restype = C.c_int; tlf.argtypes = [C.c_void_p, C.c_int]
dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
def run(M, N, K, tile, nblocks):
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev); c = torch.zeros(M, N, device=dev)
    for _ in range(3): nt(a.data_ptr(), w.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, tile, st)
    torch.cuda.synchronize()
    buf = np.zeros((nblocks, 8), dtype=np.uint64); tlf(buf.ctypes.data, nblocks)
    t = buf[:, :6].astype(np.float64); ok = t[:, 4] > 0; t = t[ok]
    per = t[:, [0, 1, 5, 2, 3]] / t[:, 4:5]
    print(f"NT {M}x{N}x{K} tile {tile}: blocks {ok.sum()}, phases/block {np.median(t[:,4]):.0f}; cycles per phase (median over blocks): "
          f"load-issue {np.median(per[:,0]):.0f}  lds-read+mfma-issue {np.median(per[:,1]):.0f}  vm-wait {np.median(per[:,2]):.0f}  split+lds-store {np.median(per[:,3]):.0f}  barrier {np.median(per[:,4]):.0f}  total {np.median(per.sum(1)):.0f}")
run(3840, 128, 1152, 19, 240)
run(3840, 128, 1152, 35, 240)
run(960, 256, 2304, 18, 240)
run(3840, 768, 256, 13, 720)
run(3840, 768, 256, 0, 720)
run(15360, 384, 128, 0, 1440)
run(61440, 192, 64, 0, 2880)
run(15360, 384, 128, 30, 1440)
