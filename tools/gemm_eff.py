"""per C-ABI op and per (GEMM instantiation, problem shape): time and TFLOP/s inside the eager bench step (HIP-event timing).

    python tools/gemm_eff.py [AMP] [--batch B]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sast_amd.profiling import gemm_report

args = [a for a in sys.argv[1:] if not a.startswith("--")]
amp = float(args[0]) if args else 2e-4
if "--batch" in sys.argv:
    bench.BATCH = int(sys.argv[sys.argv.index("--batch") + 1])
tr = bench.Trainer(torch.device("cuda:0"), amp, 1, use_graph=False)
tr.capture()


def run(n):
    for _ in range(n):
        tr.eager_step()


rows_all = gemm_report(run, 3, per_shape=True)
ops = [r for r in rows_all if r[0].startswith("op:")]
rows = [r for r in rows_all if not r[0].startswith("op:")]
print(f"op-level (C-ABI call) breakdown, total {sum(r[2] for r in ops) / 3:.3f} ms/step")
for name, n, ms, fl in ops:
    print(f"{ms / 3:8.3f} ms/step {n // 3:4d} calls {1e3 * ms / n:8.1f} us/call  {name}")
tot = sum(r[2] for r in rows)
print(f"GEMM family: {tot / 3:.3f} ms/step, {sum(r[3] for r in rows) / 3 / 1e9:.1f} GFLOP/step")
for name, n, ms, fl in rows:
    kern, _, shape = name.partition(" |")
    print(f"{ms / 3:8.3f} ms/step {n // 3:3d} calls {1e3 * ms / n:7.1f} us {fl / ms / 1e9:6.1f} TF/s  [{shape}]  {kern[:170]}")
