import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sast_amd import _lib as L
lib = L.tools_lib(); fn = lib.sast_test_gemm_nt; fn.restype = C.c_int; fn.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p]
M, N, K, t = [int(v) for v in sys.argv[1:5]]
dev = torch.device("cuda:0")
a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev); c = torch.empty(M, N, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(10):
    fn(a.data_ptr(), w.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, t, st)
torch.cuda.synchronize()
