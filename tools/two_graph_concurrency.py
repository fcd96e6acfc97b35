"""Do two SEPARATELY INSTANTIATED hipGraphs replayed on two HIP streams overlap on this stack -- and what would stage-skewed
micro-batches buy the 1Mpx B = 4 step?  (round-4 verdict item 3; round 3 measured that a fork/join captured INTO one graph serialises,
`profiles/r03_o_graph_branch_concurrency.txt`, while eager two-stream launches overlap.)

The backbone is per sample (sast_rnn.py:156-161): a B = 4 batch can run as two independent B = 2 chains.  Each chain (own module
instance = own weights, gradients and scratch: nothing shared, so whatever is measured is scheduling, not atomics) is captured as
graphs on its own stream:

  whole      one graph per chain: input prep, stages 1-4 forward, proxy loss on the stage 2-4 states, backward
  3 phases   G1 = input prep + stages 1-2 forward (HBM-bound rows), G2 = stages 3-4 forward + loss + their backward (latency-bound
             chain of small GEMMs), G3 = backward of stages 1-2

  b4            the B = 4 chain as ONE graph on one stream (what the product replays today, backbone part)
  b2 serial     chain A then chain B on ONE stream
  b2 parallel   A on stream 1, B on stream 2, launched together (phases coincide)
  b2 skewed     three-phase graphs: B's G1 waits for A's G1 -- A's latency-bound phase runs beside B's bandwidth-bound one

Wall time per pair from HIP events on the launching stream (fork / join through events), 30 repetitions after 5 warm-up."""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sast_amd import functional as SF
from sast_amd.config import backbone_config
from sast_amd.detection import RNNDetector

HW, PART = (384, 640), (6, 10)
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
FUSED_ALWAYS = "--fused" in sys.argv
if FUSED_ALWAYS:
    SF._FUSED_MIN_ROWS = 0
FWD_ONLY = "--fwd-only" in sys.argv


def events(B, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(B, 20, HW[0], HW[1], generator=g) > 0.5).int().to(dev)


class Chain:
    def __init__(self, net, x):
        self.net, self.x = net, x
        self.one = torch.ones((), device=dev)
        self.stream = torch.cuda.Stream()
        self.graphs = None
        self.net._prep_ws = {}

    def zero(self):
        for p in self.net.parameters():
            if p.grad is not None:
                p.grad.zero_()

    # ---- the phases
    def phase1(self):
        r, xin = SF.input_prep(self.x, None, self.net._prep_ws, keep_bytes=SF.STEM_U8)
        self.r = r
        self.feats = {}
        for i in (0, 1):
            xin, state, _p = self.net.stages[i].forward_nhwc(xin, None, r[:, i])
            self.feats[i + 1] = state[0]
        self.mid = xin

    def phase2(self):
        if FWD_ONLY:
            xin = self.mid
            for i in (2, 3):
                xin, state, _p = self.net.stages[i].forward_nhwc(xin, None, self.r[:, i])
                self.feats[i + 1] = state[0]
            return
        self.leaf = self.mid.detach().requires_grad_(True)
        self.f2 = self.feats[2].detach().requires_grad_(True)
        xin = self.leaf
        for i in (2, 3):
            xin, state, _p = self.net.stages[i].forward_nhwc(xin, None, self.r[:, i])
            self.feats[i + 1] = state[0]
        loss = SF.mean_squares(self.f2, self.feats[3], self.feats[4])
        loss.backward(gradient=self.one)

    def phase3(self):
        if FWD_ONLY:
            return
        torch.autograd.backward([self.mid, self.feats[2]], [self.leaf.grad, self.f2.grad])

    def all_phases(self):
        self.phase1(); self.phase2(); self.phase3()

    def warm(self):
        with torch.cuda.stream(self.stream):
            with torch.set_grad_enabled(not FWD_ONLY):
                for _ in range(2):
                    self.zero()
                    self.all_phases()
        torch.cuda.synchronize()

    def capture(self, split):
        self.warm()
        pool = torch.cuda.graph_pool_handle()
        gs = []
        with torch.set_grad_enabled(not FWD_ONLY):
            if split:
                for ph in (self.phase1, self.phase2, self.phase3):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, pool=pool, stream=self.stream, capture_error_mode="thread_local"):
                        ph()
                    gs.append(g)
            else:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool, stream=self.stream, capture_error_mode="thread_local"):
                    self.all_phases()
                gs.append(g)
        self.graphs = gs
        torch.cuda.synchronize()


def timeit(fn, reps=30):
    main = torch.cuda.current_stream()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(main)
    for _ in range(reps):
        fn()
    e1.record(main)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def make_net(amp):
    torch.manual_seed(0)
    return RNNDetector(backbone_config(HW, PART, embed_dim=64, AMP=amp)).to(dev)


def run(amp):
    main = torch.cuda.Stream()
    torch.cuda.set_stream(main)
    x4 = events(4, 0)
    base = make_net(amp)
    c4 = Chain(base, x4)
    c4.capture(split=False)
    c4s = Chain(copy.deepcopy(base), x4)
    c4s.capture(split=True)
    nets = [copy.deepcopy(base), copy.deepcopy(base)]
    whole = [Chain(nets[i], x4[2 * i:2 * i + 2].contiguous()) for i in range(2)]
    for c in whole:
        c.capture(split=False)
    nets3 = [copy.deepcopy(base), copy.deepcopy(base)]
    three = [Chain(nets3[i], x4[2 * i:2 * i + 2].contiguous()) for i in range(2)]
    for c in three:
        c.capture(split=True)

    def b4():
        c4.graphs[0].replay()

    def b4_three():
        for g in c4s.graphs:
            g.replay()

    def serial():
        for c in whole:
            c.graphs[0].replay()

    def fork(chains):
        ev = torch.cuda.Event()
        ev.record(main)
        for c in chains:
            c.stream.wait_event(ev)

    def join(chains):
        for c in chains:
            main.wait_stream(c.stream)

    def parallel():
        fork(whole)
        for c in whole:
            with torch.cuda.stream(c.stream):
                c.graphs[0].replay()
        join(whole)

    def parallel_three():
        fork(three)
        for k in range(3):
            for c in three:
                with torch.cuda.stream(c.stream):
                    c.graphs[k].replay()
        join(three)

    def skewed():
        a, b = three
        fork(three)
        with torch.cuda.stream(a.stream):
            a.graphs[0].replay()
            e = torch.cuda.Event()
            e.record(a.stream)
        b.stream.wait_event(e)
        with torch.cuda.stream(a.stream):
            a.graphs[1].replay()
        with torch.cuda.stream(b.stream):
            b.graphs[0].replay()
        with torch.cuda.stream(a.stream):
            a.graphs[2].replay()
        with torch.cuda.stream(b.stream):
            b.graphs[1].replay()
            b.graphs[2].replay()
        join(three)

    def single_b2():
        whole[0].graphs[0].replay()

    def phases_b2():
        out = []
        for k in range(3):
            out.append(timeit(lambda: three[0].graphs[k].replay()))
        return out

    res = {"b4 one graph": timeit(b4), "b4 three graphs": timeit(b4_three), "b2 alone (one chain)": timeit(single_b2),
           "b2+b2 serial, one stream": timeit(serial), "b2+b2 parallel, two streams": timeit(parallel),
           "b2+b2 parallel, three-phase graphs": timeit(parallel_three), "b2+b2 skewed by one phase": timeit(skewed)}
    ph = phases_b2()
    print(f"AMP {amp:g}  ({'forward only' if FWD_ONLY else 'forward + backward'}, backbone, 1Mpx, fused forward {'forced' if FUSED_ALWAYS else 'by policy'})")
    for k, v in res.items():
        print(f"  {k:38s} {v:7.3f} ms   {v / res['b4 one graph']:.3f} x b4")
    print(f"  phases of one b2 chain alone: G1 {ph[0]:.3f}  G2 {ph[1]:.3f}  G3 {ph[2]:.3f} ms")
    torch.cuda.set_stream(torch.cuda.default_stream())


if __name__ == "__main__":
    for amp in (2e-4, 1.0):
        run(amp)
