"""Where the HOST time of an eager step goes (the reference's own caller -- Lightning's training_step -- launches the modules eagerly;
only sast_amd.training.TrainStep replays hipGraphs).  cProfile over a few eager steps of bench.py's workload.

    python tools/eager_profile.py [--steps 20] [--fwd-bwd-only]
"""
import argparse
import cProfile
import io
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--amp", type=float, default=2e-4)
    ap.add_argument("--top", type=int, default=45)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    tr = bench.Trainer(dev, a.amp, 1, False)
    for _ in range(40):          # (the first few dozen eager steps still pay one-time costs: allocator growth, code-object loads)
        tr.eager_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        tr.eager_step()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"eager step: {1e3 * t_all / a.steps:.3f} ms wall, {1e3 * t_issue / a.steps:.3f} ms of host issue time per step")
    # how much of the issue time is spent INSIDE the C entry points (argument checks + hipLaunchKernel): the floor no Python change moves
    from sast_amd import _lib as SL
    lib, acc = SL.lib(), {}
    names = [n for n in dir(lib) if n.startswith("sast_")] + [n for n in list(vars(lib)) if n.startswith("sast_")]

    def wrap(name, fn):
        slot = acc.setdefault(name, [0, 0.0])

        def call(*args):
            t = time.perf_counter()
            r = fn(*args)
            slot[1] += time.perf_counter() - t
            slot[0] += 1
            return r
        return call
    saved = {}
    for n in set(names):
        f = getattr(lib, n)
        if callable(f):
            saved[n] = f
            setattr(lib, n, wrap(n, f))
    t0 = time.perf_counter()
    for _ in range(a.steps):
        tr.eager_step()
    t_issue2 = time.perf_counter() - t0
    torch.cuda.synchronize()
    for n, f in saved.items():
        setattr(lib, n, f)
    tot_c = sum(v[1] for v in acc.values())
    tot_n = sum(v[0] for v in acc.values())
    print(f"inside the C entry points: {1e3 * tot_c / a.steps:.3f} ms per step over {tot_n // a.steps} calls per step "
          f"({1e6 * tot_c / max(tot_n, 1):.1f} us per call; issue time of these steps {1e3 * t_issue2 / a.steps:.3f} ms)")
    for n, (k, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:12]:
        if k:
            print(f"   {n:34s} {k // a.steps:4d} calls/step  {1e6 * t / k:7.1f} us each")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(a.steps):
        tr.eager_step()
    pr.disable()
    torch.cuda.synchronize()
    for key in ("tottime", "cumulative"):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(a.top)
        print(f"---- by {key} (over {a.steps} steps; cProfile inflates every call) ----")
        print("\n".join(l[:170] for l in s.getvalue().splitlines()[4:]))


if __name__ == "__main__":
    main()
