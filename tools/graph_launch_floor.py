"""the floor of a kernel inside a replayed hipGraph: N dependent elementwise kernels on a tensor of `numel` floats, captured once,
replayed; microseconds per kernel node.  (What a launch boundary costs the 280-launch training step.)"""
import sys, torch
dev = torch.device("cuda:0")
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
for numel in (1, 4096, 1 << 20, 4 << 20, 16 << 20):
    x = torch.zeros(numel, device=dev)
    for n in (50, 200):
        for _ in range(3):
            x.add_(1.0)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(n):
                x.add_(1.0)
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20 / n
        bw = 8.0 * numel / us / 1e6
        print(f"numel {numel:9d}  {n:3d} nodes per graph: {us:6.2f} us per kernel node   ({bw:7.3f} TB/s of read+write)")
