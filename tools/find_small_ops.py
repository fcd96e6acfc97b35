"""which ATen ops (python call sites) still launch kernels in one eager step of bench.Trainer (they are plumbing around the
HIP library: autograd sums, the proxy loss, buffer clears) -- used to hunt launch-latency-bound leftovers."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
tr = bench.Trainer(dev, 2e-4, 1, use_graph=False)
for _ in range(2):
    tr.fwd_bwd(); tr.update()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    tr.fwd_bwd(); tr.update()
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name.startswith("aten::") and e.cpu_parent is None or (e.cpu_parent is not None and not e.cpu_parent.name.startswith("aten::") and e.name.startswith("aten::")):
        kern = [k.name[:60] for k in e.kernels] if hasattr(e, "kernels") else []
        if not kern:
            continue
        st = [s for s in (e.stack or []) if "/root/repo" in s or "sast_amd" in s or "bench.py" in s]
        where = st[0].split("/")[-1] if st else (e.cpu_parent.name if e.cpu_parent is not None else "autograd engine")
        cnt[(e.name, str(e.input_shapes)[:60], where[:70])] += 1
for (name, shp, where), n in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print(f"{n:4d}  {name:28s} {shp:60s} {where}")
print("---- copies / fills (any)")
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::clone", "aten::contiguous", "aten::zeros", "aten::zeros_like", "aten::full", "aten::empty_like", "aten::to"):
        st = [s for s in (e.stack or []) if ("sast_amd" in s or "bench.py" in s)]
        where = st[0].split("/")[-1] if st else (e.cpu_parent.name if e.cpu_parent is not None else "?")
        if e.name in ("aten::empty_like",):
            continue
        cnt[(e.name, str(e.input_shapes)[:50], where[:80])] += 1
for (name, shp, where), n in sorted(cnt.items(), key=lambda kv: -kv[1])[:50]:
    print(f"{n:4d}  {name:20s} {shp:50s} {where}")
print("---- memcpy activities")
tbl = prof.key_averages(group_by_stack_n=6)
for r in tbl:
    if "emcpy" in r.key or "emset" in r.key or "copy" in r.key.lower():
        print(r.count, r.key, [s.split("/")[-1] for s in (r.stack or [])][:6])
