"""Which Python lines of the step enqueue ATen kernels / device-to-device copies (not this library's kernels)?

One eager headline step under torch.profiler with stacks; every aten op that launches a device kernel or a DtoD memcpy is listed with
the innermost frames of sast_amd / bench.py that called it.  python tools/find_aten_launches.py [--batch 4]"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    a = ap.parse_args()
    bench.BATCH = a.batch
    tr = bench.Trainer(torch.device("cuda:0"), 2e-4, 1, False)
    for _ in range(3):
        tr.eager_step()
    torch.cuda.synchronize()
    # (1) device work owned by ATen ops (kernels), from the profiler
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        tr.eager_step()
        torch.cuda.synchronize()
    rows = collections.Counter()
    for ev in prof.events():
        if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith("aten::"):
            continue
        kern = list(ev.kernels)
        if not kern or any(c.kernels for c in ev.cpu_children):      # only the op that DIRECTLY owns the device work
            continue
        for k in kern:
            rows[(ev.name, k.name[:60])] += 1
    print("ATen ops that launch kernels in one eager step:")
    for (op, k), n in sorted(rows.items(), key=lambda kv: -kv[1]):
        print(f"{n:3d}  {op:24s} {k}")
    mem = collections.Counter(ev.name for ev in prof.events() if "Memcpy" in ev.name or "Memset" in ev.name)
    print("runtime copies / memsets:", dict(mem))

    # (2) who asks for copies: Tensor.clone / contiguous / copy_ / to wrapped at the Python level (the autograd thread included)
    import traceback
    sites = collections.Counter()

    def site():
        fr = [f for f in traceback.extract_stack()[:-2] if "/sast_amd/" in f.filename or f.filename.endswith("bench.py")]
        return " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(fr[-3:])) or "(outside the repo)"

    def wrap(name, copies):
        orig = getattr(torch.Tensor, name)

        def f(self, *a, **k):
            if self.is_cuda and copies(self, a, k):
                sites[(name, tuple(self.shape), site())] += 1
            return orig(self, *a, **k)
        setattr(torch.Tensor, name, f)
        return orig

    saved = {n: wrap(n, c) for n, c in (("clone", lambda s, a, k: True), ("contiguous", lambda s, a, k: not s.is_contiguous()),
                                        ("copy_", lambda s, a, k: True), ("to", lambda s, a, k: True), ("float", lambda s, a, k: s.dtype != torch.float32))}
    try:
        tr.eager_step()
        torch.cuda.synchronize()
    finally:
        for n, o in saved.items():
            setattr(torch.Tensor, n, o)
    print("Python-level copy requests in one eager step:")
    for (name, shape, where), n in sorted(sites.items(), key=lambda kv: -kv[1]):
        print(f"{n:3d}  {name:10s} {str(shape):28s} {where}")

if __name__ == "__main__":
    main()
