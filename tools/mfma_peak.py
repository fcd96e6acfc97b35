"""v_mfma_f32_32x32x2_f32 issue-rate calibration on the MI355X: TFLOP/s a workgroup of 4 waves (one per SIMD) sustains with the
loop structure of the GEMM template, at 1 / 2 / 3 / 4 workgroups per CU.  Peak: 157.3 TFLOP/s (256 CUs x 4 SIMDs x 64 flop/cycle x 2.4 GHz).
modes: 0 dependent chain on one accumulator, 1 two accumulators, 2 chain + operands re-read from LDS per 8 MFMAs,
       3 chain + workgroup barrier per 8 MFMAs, 4 LDS reads + barrier (the k-loop of gemm_body without the global loads)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sast_amd import _lib as L
lib = L.tools_lib()
fn = lib.sast_test_mfma_peak; fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
out = torch.zeros(16, device="cuda"); st = torch.cuda.current_stream().cuda_stream
iters = 4000
names = {0: "chain", 1: "2 accs", 2: "chain+LDS", 3: "chain+barrier", 4: "LDS+barrier"}
print("blocks/CU " + " ".join(n.rjust(14) for n in names.values()))
for per_cu in (1, 2, 3, 4):
    line = f"{per_cu:9d} "
    for mode in names:
        blocks = 256 * per_cu
        fn(out.data_ptr(), mode, blocks, 10, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(out.data_ptr(), mode, blocks, iters, st); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        fl = blocks * 4 * iters * 8 * 2.0 * 32 * 32 * 2
        line += f"{fl / ms / 1e9:11.1f} TF"
    print(line)
