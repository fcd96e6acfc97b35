"""float4-copy calibration of the box this call runs on (round 6, verdict item 3): achieved HBM rate (bytes read + written / time) of a
streaming elementwise pass over buffers far larger than the 256 MB Infinity Cache -- what "the HBM roof" is in practice for the tables
of profiles/ (guide: 6.29 TB/s for the same pattern; spec 8 TB/s).  Prints one JSON line."""
import json
import torch

dev = torch.device("cuda:0")
n = 1 << 29                                   # 2 GiB per fp32 buffer
x = torch.empty(n, device=dev).uniform_()
y = torch.empty_like(x)
out = {}
for name, fn in (("elementwise_float4 (torch.add(x, 0, out=y): vectorized_elementwise_kernel, 16 B per lane)", lambda: torch.add(x, 0.0, out=y)),
                 ("hipMemcpyAsync D2D (y.copy_(x))", lambda: y.copy_(x))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    out[name] = {"ms": ms, "tb_per_s": 2 * 4 * n / (ms * 1e-3) / 1e12}
out["copy_tbs"] = max(v["tb_per_s"] for v in out.values())
print(json.dumps(out))
