"""The training step of the hot path: forward, backward in SEGMENTS, bucketed gradient all-reduce overlapped with the
remaining backward, fused AdamW per bucket.

Reference: Lightning's DDP step around Module.training_step (modules/detection.py:113-221), `DDPStrategy(gradient_as_bucket_view=
True)` (train.py:96-98) -- DDP all-reduces gradient buckets while the backward pass is still running.  Here the same overlap is
explicit and hipGraph-friendly:

  bucket 0  PAFPN (+ YOLOX head)      final after segment A  (loss -> FPN inputs)
  bucket 1  backbone stage 4          final after segment B  (stage-4 backward)
  bucket 2  backbone stages 3, 2, 1   final after segment C

The parameters / gradients of a bucket are one contiguous slice of the flat buffers (dist.FlatParams).  After each segment the
main stream records an event; a side stream waits for it, all-reduces that bucket over RCCL (torch.distributed "nccl") and runs
the AdamW update of the bucket, while the main stream continues with the next segment (the later segments never read the
parameters of an earlier bucket again).  The next step's forward waits for the side stream.  Each segment can be a captured
hipGraph (three graphs sharing one memory pool); the GRADIENT collectives stay outside the graphs (they run on the side stream
between them).  The only captured RCCL calls are SyncBatchNorm's statistics all-reduces (functional.SyncBatchNormGroup), which sit
in the middle of the PAFPN / head passes.
With world == 1 the all-reduce is a no-op and everything else is identical, so a single-GPU run exercises the same path.

For sequences (seq_len > 1, BPTT) the backbone gradients only become final at the end of the backward: segments B and C merge.

Round 6: DEFERRED WEIGHT GRADIENTS (`defer_dw=True`).  Nothing reads a weight gradient before AdamW, yet in the chain above every
layer's backward launch carries its dW job beside its dX job and the next kernel waits for both.  With deferral the backward entry
points launch only the dX chain on the main stream and park the dW jobs in the library (include/sast_hip.h: sast_dw_defer); after each
segment the side stream waits for the segment's event, runs the parked jobs (`SF.dw_flush`; replayed as a hipGraph of its own -- kernel
nodes of ONE graph never overlap on ROCm 7.2, separately instantiated graphs on two streams do) and then that segment's all-reduce +
AdamW, while the main stream is already in the next segment's chain.  `dw_cuts` adds segment boundaries before backbone stages
(more, shorter dW graphs: less weight-gradient work left exposed behind the last segment); buckets and their order are unchanged.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence

import torch
import torch.distributed as dist

from . import functional as SF
from .detection.network_blocks import pass_sync_group
from .dist import FlatParams, FusedAdamW, OneCycleLR


class TrainStep:
    def __init__(self, net, fpn, head=None, *, lr: float = 2e-4, weight_decay: float = 0.0, clip_value: float = 0.0, eps: float = 1e-8,
                 schedule: Optional[OneCycleLR] = None, world: int = 1, group=None, segmented: bool = True, defer_dw: bool = False,
                 dw_rows=(0, 0), dw_discard: bool = False, cuts: Sequence[int] = (3,)):
        self.net, self.fpn, self.head = net, fpn, head
        self.world, self.group = world, group
        # defer_dw: weight gradients off the backward chain (module docstring); dw_rows = (min, max) reduction rows of the jobs that are
        # parked (0 = unbounded); dw_discard: TIMING PROBE ONLY -- the parked jobs are dropped instead of run (the dX chain alone)
        self.defer_dw, self.dw_rows, self.dw_discard = bool(defer_dw), tuple(dw_rows), bool(dw_discard)
        stages = list(net.stages)
        first = [fpn] + ([head] if head is not None else [])
        # The segmented backward is written for the topology of the shipped models: four backbone stages, the PAFPN reading stages up to
        # the LAST one (its top feature map is the stage-4 state: segment B starts there), and no parameter shared between buckets --
        # the side stream's AdamW rewrites a bucket's weights while the main stream still runs the backward of the later buckets, so a
        # later segment must never read (or add gradient to) a parameter of an already-updated bucket.  Anything else runs unsegmented.
        if segmented:
            why = None
            if len(stages) != 4:
                why = f"{len(stages)} backbone stages (4 expected)"
            elif list(fpn.in_features)[-1] != len(stages):
                why = f"PAFPN in_stages {tuple(fpn.in_features)} do not end at the last backbone stage"
            else:
                owner = {}
                for b, grp in enumerate([first, [stages[3]], stages[:3]]):
                    for m in grp:
                        for prm in m.parameters():
                            if owner.setdefault(id(prm), b) != b:
                                why = "a parameter is shared between gradient buckets"
            if why is not None:
                import warnings
                warnings.warn(f"sast_amd.TrainStep: segmented backward disabled ({why}); running the monolithic step")
                segmented = False
        self.segmented = segmented
        # cuts: the backbone stages (0-based) whose INPUT is a segment boundary of the single-timestep backward.  (3,) = the three
        # segments of the module docstring; more cuts (e.g. (3, 2, 1)) only add boundaries -- the buckets stay the same three, a bucket
        # is reduced + updated behind the segment that completes it; with defer_dw every segment's parked weight gradients leave behind it
        self.cuts = tuple(sorted({int(c) for c in cuts}, reverse=True))
        if segmented and (not self.cuts or self.cuts[0] != 3 or min(self.cuts) < 1):
            raise ValueError(f"TrainStep(cuts={cuts}): the segmented backward needs the cut before the last stage (3) and cuts in 1 .. 3")
        if len(stages) == 4:
            buckets = [first, [stages[3]], [stages[2], stages[1], stages[0]]]
        else:
            buckets = [first, list(reversed(stages))]
        self.flat = FlatParams([], buckets=buckets)
        self.opt = FusedAdamW(self.flat, lr=lr, weight_decay=weight_decay, clip_value=clip_value, eps=eps, schedule=schedule)
        dev = self.flat.flat.device
        self._one = torch.ones((), device=dev)
        # the side stream carries what the step does NOT wait for (bucket all-reduce + AdamW, deferred weight gradients): lowest priority
        # with defer_dw, so that the dispatcher serves the backward chain's workgroups first and the parked jobs soak up what is left
        prio = int(os.environ.get("SAST_SIDE_PRIORITY", "1" if defer_dw else "0"))
        self.side = torch.cuda.Stream(priority=prio) if dev.type == "cuda" else None
        self.loss = self.P = self.losses = None
        self._graphs = None
        self._dw_graphs = None
        self._seg_state = None
        # measure_exposed: every step records (main stream idle, side stream done) event pairs -- how long the step waits for the
        # last bucket's all-reduce + update AFTER its own backward has finished (bench.py: `allreduce_exposed_ms`)
        self.measure_exposed = False
        self._exposed = []
        self._bucket_ev = {}       # bucket -> [(start, end)] event pairs on the stream that runs its all-reduce + update (measure_exposed)

    # ---------------------------------------------------------------- forward + backward segments
    def forward(self, xs: Sequence[torch.Tensor], states=None, labels=None, indices=None, token_masks=None):
        """xs: the event tensors of the sequence (one timestep = BASELINE's metric).  labels None: proxy objective
        sum_k mean(out_k^2) on the PAFPN outputs of the last timestep; else the YOLOX / SimOTA loss on `labels`, with
        `indices` (per timestep the batch indices that carry labels, modules/detection.py:161-171) or on the last timestep."""
        if self.defer_dw and SF.dw_pending():
            # a backward of an earlier step died between parking and flushing: its jobs point at buffers that are gone -- never run them
            SF.dw_discard()
            SF.dw_release()
        self.flat.zero_grad()
        single = len(xs) == 1 and self.segmented
        feats_seq, Ps = [], []
        for t, x in enumerate(xs):
            feats, states, P = self.net.forward_nhwc(x, states, token_masks[t] if token_masks is not None else None,
                                                     cut_before_stage=self.cuts if single else None)
            feats_seq.append(feats)
            Ps.append(P)
        cut = dict(self.net.last_cuts) if (single and self.net.last_cuts) else None
        if indices is not None:
            from .detection.sequence import BackboneFeatureSelector
            sel = BackboneFeatureSelector()
            for f, idx in zip(feats_seq, indices):
                if idx is not None and len(idx) > 0:
                    sel.add_backbone_features({k: f[k] for k in self.fpn.in_features}, idx)
            fpn_in = sel.get_batched_backbone_features()
        else:
            fpn_in = {k: feats_seq[-1][k] for k in self.fpn.in_features}
        leaves = None
        if self.segmented:      # FPN inputs as detached leaves: segment A ends there
            leaves = {k: v.detach().requires_grad_(True) for k, v in fpn_in.items()}
        outs = self.fpn.forward_nhwc(leaves if leaves is not None else fpn_in)
        if labels is not None:
            _pred, self.losses = self.head.forward_train_nhwc(outs, labels)
            loss = self.losses["loss"]
        else:
            loss = SF.mean_squares(*outs)
        self.loss, self.P, self.states = loss, Ps[-1], states
        self._seg_state = (loss, fpn_in, leaves, cut)
        return loss

    def _stage_groups(self):
        """single-timestep segmented backward: the backbone stages (0-based) of segments 1, 2, ...: [[3], [2, 1, 0]] for cuts (3,)"""
        cut = self._seg_state[3]
        bounds = sorted(cut.keys(), reverse=True)           # e.g. [3, 2, 1]
        groups, hi = [], len(self.net.stages) - 1
        for c in bounds:
            groups.append(list(range(hi, c - 1, -1)))
            hi = c - 1
        groups.append(list(range(hi, -1, -1)))
        return groups

    def n_segments(self) -> int:
        if not self.segmented:
            return 1
        return 1 + len(self._stage_groups()) if self._seg_state[3] is not None else 2

    def _side_path(self) -> bool:
        """reduce + update (and the parked weight-gradient jobs) run on the side stream"""
        return self.side is not None and (self.segmented or self.defer_dw)

    def backward_segment(self, i: int):
        if self.defer_dw:
            prev = SF.dw_defer(True, *self.dw_rows)
            try:
                self._backward_segment(i)
            finally:
                SF.dw_defer(prev)
            return
        self._backward_segment(i)

    def _backward_segment(self, i: int):
        loss, fpn_in, leaves, cut = self._seg_state
        if i == 0:
            loss.backward(gradient=self._one)            # a resident 1.0 instead of a ones_like fill launch per step
            return
        keys = list(fpn_in.keys())
        if cut is None:                                  # one backbone segment (sequences, or stage input without a graph)
            torch.autograd.backward([fpn_in[k] for k in keys], [leaves[k].grad for k in keys])
            return
        # segment i >= 1 runs the backward of a group of stages: its roots are the input of the stage above the group (the cut's
        # upstream tensor, gradient = what the segment before left in the cut's leaf) and the group's own feature maps the FPN reads
        grp = self._stage_groups()[i - 1]
        roots, grads = [], []
        above = max(grp) + 1
        if above in cut:
            roots.append(cut[above][0])
            grads.append(cut[above][1].grad)
        for k in keys:                                   # FPN input key k = output of stage k - 1
            if k - 1 in grp:
                roots.append(fpn_in[k])
                grads.append(leaves[k].grad)
        torch.autograd.backward(roots, grads)

    def bucket_of_segment(self, i: int) -> List[int]:
        """the gradient buckets that are FINAL behind segment i"""
        n = self.n_segments()
        if n == 1:
            return list(range(len(self.flat.bucket_ranges)))
        if n == 2:
            return [[0], [1, 2]][i]
        if i == 0:
            return [0]
        groups = self._stage_groups()
        done = {st for g in groups[:i] for st in g}
        before = {st for g in groups[:i - 1] for st in g}
        out = []
        for b, stages in ((1, {3}), (2, {2, 1, 0})):     # bucket 1 = stage 4, bucket 2 = stages 3, 2, 1 (FlatParams order of __init__)
            if stages <= done and not stages <= before:
                out.append(b)
        return out

    # ---------------------------------------------------------------- reduce + update
    def _reduce_update(self, buckets: List[int]):
        for b in buckets:
            if self.measure_exposed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            self.flat.all_reduce(self.group, bucket=b)
            self.opt.update(grad_scale=1.0 / self.world, bucket=b)
            if self.measure_exposed:
                e1.record()
                self._bucket_ev.setdefault(b, []).append((e0, e1))

    def bucket_ms(self):
        """mean duration of each bucket's all-reduce + AdamW on the stream that ran it (measure_exposed steps): with the segmented step
        all but the last run beside the remaining backward, so their sum minus `exposed_ms` is what the overlap hid"""
        torch.cuda.synchronize()
        out = {b: sum(a.elapsed_time(e) for a, e in v) / len(v) for b, v in sorted(self._bucket_ev.items()) if v}
        self._bucket_ev = {}
        return out

    def _flush_dw(self, i: int):
        """the weight-gradient jobs segment i parked, on the current (side) stream: the captured graph of the flush, or the launches"""
        if not self.defer_dw:
            return
        if self._dw_graphs is not None:
            if self._dw_graphs[i] is not None:
                self._dw_graphs[i].replay()
        elif self.dw_discard:
            SF.dw_discard()
        else:
            SF.dw_flush()

    def flush_pending(self):
        """callers that run forward + backward_segment() themselves (no reduce / update): the parked jobs on the CURRENT stream"""
        if self.defer_dw:
            SF.dw_flush()
            SF.dw_release()

    def _after_segment(self, i: int, first: bool):
        """(the parked weight-gradient jobs of segment i, then) all-reduce + AdamW of the buckets that segment i completed; on the side
        stream when the step is segmented or defers its weight gradients"""
        if not self._side_path():
            if self.defer_dw:
                self._flush_dw(i)
            if first:
                self.opt.begin_step()
            self._reduce_update(self.bucket_of_segment(i))
            return
        main = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(main)
        with torch.cuda.stream(self.side):
            self.side.wait_event(ev)
            self._flush_dw(i)
            if first:
                self.opt.begin_step()
            self._reduce_update(self.bucket_of_segment(i))

    def finish(self):
        if self._side_path():
            main = torch.cuda.current_stream()
            if self.measure_exposed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(main)            # the main stream has nothing left of this step ...
                e1.record(self.side)       # ... the side stream still owes the last bucket's reduce + update
                self._exposed.append((e0, e1))
            main.wait_stream(self.side)
        if self.defer_dw and self._graphs is None:
            SF.dw_release()          # eager: the allocating stream is now ordered behind the flushed launches

    def exposed_ms(self):
        """mean time per step by which the side stream (bucket all-reduce + AdamW) finished AFTER the main stream (0 when it was
        done first); for the unsegmented step with world > 1 the whole reduce + update is exposed and bracketed directly"""
        torch.cuda.synchronize()
        if not self._exposed:
            return None
        v = [max(0.0, a.elapsed_time(b)) for a, b in self._exposed]
        self._exposed = []
        return sum(v) / len(v)

    # ---------------------------------------------------------------- eager step
    def step(self, xs, states=None, labels=None, indices=None, token_masks=None):
        self.flat.check_views()
        self.forward(xs, states, labels, indices, token_masks)
        for i in range(self.n_segments()):
            self.backward_segment(i)
            self._after_segment(i, first=(i == 0))
        self.finish()
        return self.loss

    # ---------------------------------------------------------------- hipGraph: one graph per segment, collectives outside
    def capture(self, xs, states=None, labels=None, indices=None, token_masks=None):
        """capture the step as hipGraphs (the inputs must be static tensors); `replay()` then runs a whole step.
        segmented: forward + segment A, segment B, segment C as graphs sharing one memory pool, reduce + update between them
        (never captured).  not segmented and world == 1: ONE graph including the optimizer update; not segmented and
        world > 1: one graph for forward + backward, all-reduce + update behind it."""
        self.flat.check_views()
        if SF.AUTOGRAD_GRADS:
            # with gradients returned on the autograd edges the parameters' AccumulateGrad nodes add them into the flat views -- on the
            # stream each node was created on (an eager step before the capture), not on the capture stream: measured garbage in the
            # segmented graphs (round 5).  The in-place contract has no AccumulateGrad work at all.
            raise RuntimeError("sast_amd.TrainStep.capture needs the in-place gradient mode (functional.set_autograd_visible_grads(False)): "
                               "the autograd-visible mode is for callers that own the backward (DDP, torch.autograd.grad), not for the "
                               "captured flat-buffer step")
        for m in (self.fpn, self.head):
            grp = pass_sync_group(m) if m is not None else None
            if grp is not None and grp.active() and not grp.capturable():
                raise RuntimeError("sast_amd.TrainStep.capture: the model was converted with convert_sync_batchnorm and the process group's "
                                   f"backend ({dist.get_backend(grp.group)}) runs its collectives on the host; the statistics all-reduces "
                                   "inside the PAFPN / head passes can only be captured into hipGraphs on RCCL (\"nccl\") -- run step()")
        if not self._side_path():
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                self.forward(xs, states, labels, indices, token_masks)
                self.backward_segment(0)
                if self.world == 1:
                    self._after_segment(0, first=True)
            self._graphs = [g]
            self.loss = self.loss.detach()
            return
        pool = torch.cuda.graph_pool_handle()
        graphs, dw_graphs = [], []

        def flush_graph():
            """the jobs the segment just captured has parked, as a graph of their own (replayed on the side stream).  Everything they
            read stays referenced (functional._DW_HOLD) until ALL graphs are captured: no later capture is handed their memory."""
            if not self.defer_dw:
                return None
            if self.dw_discard:
                SF.dw_discard()
                return None
            if SF.dw_pending() == 0:
                return None
            d = torch.cuda.CUDAGraph()
            with torch.cuda.graph(d, pool=pool, capture_error_mode="thread_local"):
                SF.dw_flush()
            return d

        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
            self.forward(xs, states, labels, indices, token_masks)
            self.backward_segment(0)
        graphs.append(g)
        dw_graphs.append(flush_graph())
        for i in range(1, self.n_segments()):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
                self.backward_segment(i)
            graphs.append(g)
            dw_graphs.append(flush_graph())
        self._graphs = graphs
        self._dw_graphs = dw_graphs if self.defer_dw else None
        SF.dw_release()
        self.loss = self.loss.detach()

    def replay(self):
        if not self._side_path():
            self._graphs[0].replay()
            if self.world > 1:
                if self.measure_exposed:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                self._after_segment(0, first=True)
                if self.measure_exposed:
                    e1.record()
                    self._exposed.append((e0, e1))
            return self.loss
        for i, g in enumerate(self._graphs):
            g.replay()
            self._after_segment(i, first=(i == 0))
        self.finish()
        return self.loss
