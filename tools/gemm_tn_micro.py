"""weight-gradient (TN, split-R) GEMM micro-benchmark: tile / split / atomic-epilogue sweep."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
cfgs = [(1, 384), (3, 256), (3, 512), (4, 256), (5, 256), (6, 256), (7, 256), (8, 384)]
TSZ = {0: (64, 64), 1: (64, 64), 2: (64, 64), 3: (128, 128), 9: (128, 128), 4: (192, 64), 5: (320, 64), 6: (64, 192), 7: (192, 128), 8: (128, 64)}
print("shape (Mo,NJ,R)".ljust(22) + " ".join(f"{TSZ[t][0]}x{TSZ[t][1]}{'k2' if t in (1, 9) else ''}/{s}".rjust(13) for t, s in cfgs))
for (Mo, NJ, R) in shapes:
    dy = torch.randn(R, Mo, device=dev); x = torch.randn(R, NJ, device=dev); out = torch.zeros(Mo, NJ, device=dev); cs = torch.zeros(Mo, device=dev)
    ref = dy.t() @ x
    line = f"{Mo}x{NJ}x{R}".ljust(22)
    def run(t, total, null):
        bm, bn = TSZ[t]
        nb = ((Mo + bm - 1) // bm) * ((NJ + bn - 1) // bn)
        splits = max(1, min((total + nb - 1) // nb, (R + 255) // 256))
        out.zero_(); cs.zero_()
        rc = fn(dy.data_ptr(), x.data_ptr(), out.data_ptr(), cs.data_ptr(), Mo, NJ, R, t, splits, null, st)
        assert rc == 0
        if not null:
            err = float((out - ref).abs().max() / ref.abs().max())
            assert err < 1e-4, err
            assert float((cs - dy.sum(0)).abs().max()) < 1e-2 * float(dy.sum(0).abs().max() + 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn(dy.data_ptr(), x.data_ptr(), out.data_ptr(), cs.data_ptr(), Mo, NJ, R, t, splits, null, st)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / 20

    for t, total in cfgs:
        us = run(t, total, 0)
        line += f" {us:7.1f}us{2.0 * Mo * NJ * R / us / 1e6:4.0f}"

    print(line)
