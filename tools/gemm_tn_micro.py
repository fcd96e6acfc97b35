"""weight-gradient (TN, split-R) GEMM micro-benchmark: tile / split / atomic-epilogue sweep."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sast_amd import _lib as L
lib = L.lib()
fn = lib.sast_test_gemm_tn
fn.restype = C.c_int
fn.argtypes = [C.c_void_p] * 4 + [C.c_int] * 6 + [C.c_void_p]
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
shapes = [(192, 64, 61440), (64, 64, 61440), (320, 64, 61440), (64, 160, 61440), (384, 128, 15360), (640, 128, 15360), (128, 320, 15360),
          (768, 256, 3840), (1344, 256, 3840), (256, 672, 3840), (1536, 512, 960), (512, 512, 960), (2688, 512, 960), (512, 1344, 960)]
cfgs = [(0, 768), (0, 256), (0, 96), (1, 384), (1, 160), (2, 320), (2, 96), (2, 32)]
print("shape (Mo,NJ,R)".ljust(22) + " ".join(f"t{t}/tot{s:<4}".rjust(12) for t, s in cfgs) + "   | null-ep t0/768  t2/320")
for (Mo, NJ, R) in shapes:
    dy = torch.randn(R, Mo, device=dev); x = torch.randn(R, NJ, device=dev); out = torch.zeros(Mo, NJ, device=dev); cs = torch.zeros(Mo, device=dev)
    ref = dy.t() @ x
    line = f"{Mo}x{NJ}x{R}".ljust(22)
    nb = ((Mo + 63) // 64) * ((NJ + 63) // 64)

    def run(t, total, null):
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
    line += f"   | {run(0, 768, 1):7.1f}us {run(2, 320, 1):7.1f}us"
    print(line)
