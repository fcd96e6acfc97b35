"""print per-GEMM-instantiation time / TFLOP/s for the bench workload (HIP-event timing, eager)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sast_amd.profiling import gemm_report

amp = float(sys.argv[1]) if len(sys.argv) > 1 else 2e-4
tr = bench.Trainer(torch.device("cuda:0"), amp, 1, use_graph=False)
tr.capture()


def run(n):
    for _ in range(n):
        tr.fwd_bwd(); tr.update()


rows_all = gemm_report(run, 3)
ops = [r for r in rows_all if r[0].startswith("op:")]
rows = [r for r in rows_all if not r[0].startswith("op:")]
print(f"op-level (C-ABI call) breakdown, total {sum(r[2] for r in ops) / 3:.3f} ms/step")
for name, n, ms, fl in ops:
    print(f"{ms / 3:8.3f} ms/step {n // 3:4d} calls {1e3 * ms / n:8.1f} us/call  {name}")
tot = sum(r[2] for r in rows)
print(f"GEMM family: {tot / 3:.3f} ms/step, {sum(r[3] for r in rows) / 3 / 1e9:.1f} GFLOP/step")
for name, n, ms, fl in rows:
    print(f"{ms / 3:8.3f} ms/step {n // 3:4d} calls {1e3 * ms / n:8.1f} us/call {fl / ms / 1e9:7.1f} TF/s  {name}")
