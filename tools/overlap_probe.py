"""Do the matrix pipe and the VALU of one SIMD overlap on the MI355X?  Two waves per SIMD; cycles per loop iteration (6 MFMA
32x32x16 bf16 = 192 pipe cycles; NV plain VALU instructions) for: MFMA only, VALU only, one wave of each per SIMD, and every wave
alternating both (the instruction shape of the fused layer kernels)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sast_amd import _lib as L
lib = L.tools_lib()
fn = lib.sast_test_overlap_probe; fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
st = torch.cuda.current_stream().cuda_stream
out = torch.zeros(16, device="cuda"); cyc = torch.zeros(256, dtype=torch.int64, device="cuda")
iters = 2000
names = {0: "MFMA only (both waves)", 1: "VALU only (both waves)", 2: "one MFMA wave + one VALU wave per SIMD", 3: "each wave: 6 MFMA then NV VALU",
         4: "each wave: 2 MFMA chains then NV VALU"}
for nv in (24, 48):
    for mode, name in names.items():
        fn(out.data_ptr(), cyc.data_ptr(), mode, nv, 256, 10, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(out.data_ptr(), cyc.data_ptr(), mode, nv, 256, iters, st); e1.record(); torch.cuda.synchronize()
        c = cyc.float().mean().item() / iters
        print(f"NV {nv:3d}  {name:42s} {c:8.1f} clock64 cycles / iteration of one wave   ({e0.elapsed_time(e1) * 1e6 / iters:7.1f} ns)")
