import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import torch
from sast_amd import _lib as L
lib = L.tools_lib()
fn = lib.sast_test_gemm_tn
fn.restype = C.c_int
fn.argtypes = [C.c_void_p] * 4 + [C.c_int] * 6 + [C.c_void_p]
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
shapes = [(192, 64, 61440), (64, 64, 61440), (320, 64, 61440), (64, 160, 61440), (384, 128, 15360), (640, 128, 15360), (128, 320, 15360),
          (768, 256, 3840), (1344, 256, 3840), (256, 672, 3840), (1536, 512, 960), (512, 512, 960), (2688, 512, 960), (512, 1344, 960)]
for (Mo, NJ, R) in shapes:
    dy = torch.randn(R, Mo, device=dev); x = torch.randn(R, NJ, device=dev); out = torch.zeros(Mo, NJ, device=dev); cs = torch.zeros(Mo, device=dev)
    line = f"{Mo}x{NJ}x{R}".ljust(22)
    for total in (384, 768):
      for null in (0, 1):
        nb = ((Mo + 63) // 64) * ((NJ + 63) // 64)
        splits = max(1, min((total + nb - 1) // nb, (R + 255) // 256))
        for _ in range(3): fn(dy.data_ptr(), x.data_ptr(), out.data_ptr(), cs.data_ptr(), Mo, NJ, R, 1, splits, null, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn(dy.data_ptr(), x.data_ptr(), out.data_ptr(), cs.data_ptr(), Mo, NJ, R, 1, splits, null, st)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 50
        line += f" | b{total} {'null' if null else 'atom'} s{splits:3d} {us:6.1f}us {2.0 * Mo * NJ * R / us / 1e6:4.0f}TF {(Mo+NJ)*R*4/us/1e6:5.2f}TB/s"
    print(line)
