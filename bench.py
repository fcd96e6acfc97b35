"""bench.py -- frames/sec of the SAST hot path on MI355X.

Workload (BASELINE.json configs[2] / SURVEY §8d C3): 1Mpx 640x360 padded to 384x640, B=4 per GPU, full
SAST backbone (4 stages: conv-downsample+LN, STP scoring/selection, 2x MS-WSA, ConvLSTM) + YOLOX PAFPN,
forward + backward of the proxy loss sum_k mean(out_k^2) + gradient all-reduce (N>1) + AdamW update.
One "step" = one such pass over one synthetic batch (benchmark.py:52-64 input protocol).

    python bench.py --gpus 1 --steps 300 --warmup 50        (the defaults: the reference's benchmark.py protocol)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HW, PART, BATCH = (384, 640), (6, 10), 4
SPARSITY = 0.5


def ref_cfg(hw, part, amp=2e-4, ls=1e-5):
    from sast_amd.config import backbone_config
    return backbone_config(hw, part, embed_dim=64, AMP=amp, ls_init_value=ls)


def synthetic_events(B, hw, seed):
    """benchmark.py:58-60 protocol: (rand > sparsity).int() at the already-padded size."""
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(B, 20, hw[0], hw[1], generator=g) > SPARSITY).int()


def synthetic_labels(B, hw, num_classes, max_labels=16, seed=0):
    """(B, max_labels, 5) = (class, cx, cy, w, h) in input pixels: a random number of boxes per sample, zero rows after them
    (the layout ObjectLabels.get_labels_as_batched_tensor(format_='yolox') hands to the head, modules/detection.py:174-176)."""
    g = torch.Generator().manual_seed(1234 + seed)
    lab = torch.zeros(B, max_labels, 5)
    for b in range(B):
        n = int(torch.randint(1, max_labels + 1, (1,), generator=g))
        wh = torch.rand(n, 2, generator=g) * torch.tensor([hw[1], hw[0]]) * 0.3 + 8.0
        c = torch.rand(n, 2, generator=g) * (torch.tensor([hw[1], hw[0]]) - wh) + wh / 2
        lab[b, :n, 0] = torch.randint(0, num_classes, (n,), generator=g).float()
        lab[b, :n, 1:3], lab[b, :n, 3:5] = c, wh
    return lab


def _flush_c_stdio():
    """RCCL prints a version banner through C stdio when its first communicator comes up; on a pipe that buffer is only written at
    exit, i.e. BEHIND the JSON line.  Flushed after the process group is up (every rank) and again before rank 0 prints its line."""
    import ctypes
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)


class Trainer:
    """bench harness around sast_amd.training.TrainStep (the reference's step is Lightning's, modules/detection.py:113-221)."""

    def __init__(self, dev, amp, world, use_graph, seq_len=1, fwd_only=False, infer=False, yolox_loss=False, segmented=None,
                 label_every=0, sync_bn=False, event_dtype="int32", defer_dw=False, dw_rows=(0, 0), dw_discard=False, cuts=(3,)):
        from sast_amd.detection import RNNDetector, YOLOPAFPN
        from sast_amd.training import TrainStep
        torch.manual_seed(0)  # random init on the host, identical on every rank (weights never depend on device RNG)
        self.net = RNNDetector(ref_cfg(HW, PART, amp))
        self.fpn = YOLOPAFPN(depth=0.67, in_stages=(2, 3, 4), in_channels=(128, 256, 512))
        # host copy of the initial weights under the reference's state_dict names: the CPU baseline leg runs on them
        self.init_state = {"net": {k: v.clone() for k, v in self.net.state_dict().items() if "sub_layers" not in k},
                           "fpn": {k: v.clone() for k, v in self.fpn.state_dict().items()}}
        self.infer = infer       # --infer: backbone + PAFPN + YOLOX head in eval mode -> decoded predictions (validation.py's model part)
        self.head = None
        if infer:
            from sast_amd.detection import YOLOXHead
            self.head = YOLOXHead(num_classes=3, strides=(8, 16, 32), in_channels=(128, 256, 512)).to(dev).eval()
            self.fpn.eval()
            fwd_only = True
        self.yolox_loss = yolox_loss   # --loss yolox: the real training objective (YOLOX head + SimOTA loss) instead of the proxy loss
        rank = dist.get_rank() if world > 1 else 0
        self.labels = self.indices = None
        if yolox_loss:
            from sast_amd.detection import YOLOXHead
            self.head = YOLOXHead(num_classes=3, strides=(8, 16, 32), in_channels=(128, 256, 512)).to(dev).train()
            # --label-every k (the reference's label-sparse step, modules/detection.py:161-177): timestep t carries labels for the
            # samples b with (t + b) % k == 0, the last timestep for all; their features are gathered and batched for ONE head call
            if label_every > 0:
                self.indices = [[b for b in range(BATCH) if (t + b) % label_every == 0 or t == seq_len - 1] for t in range(seq_len)]
            n_lab = sum(len(i) for i in self.indices) if self.indices is not None else BATCH
            self.labels = synthetic_labels(n_lab, HW, 3, max_labels=16, seed=rank).to(dev)
        self.net.to(dev)
        self.fpn.to(dev)
        self.world = world
        # --sync-bn: the reference's DDP runs use SyncBatchNorm (train.py:167).  The statistics all-reduces sit between the two phases of
        # every conv + BatchNorm unit; on RCCL they are captured into the step's hipGraphs, on a host-side backend (gloo) the step runs
        # eagerly (training.TrainStep.capture refuses).  SAST_SYNC_BN_FORCE=1: the same two-phase path with ONE rank (single-GPU check)
        force = os.environ.get("SAST_SYNC_BN_FORCE", "0") == "1"
        self.sync_bn = bool(sync_bn) and (world > 1 or force) and not infer
        if self.sync_bn:
            from sast_amd.detection import convert_sync_batchnorm
            # one group for PAFPN + head: the head takes over the PAFPN's sample-count exchange of the pass
            convert_sync_batchnorm(torch.nn.ModuleList([self.fpn] + ([self.head] if self.head is not None else [])), force=force)
            if not self.fpn._sync_group.capturable():
                use_graph = False
        self.segmented = ((world > 1) and os.environ.get("SAST_SEGMENTED", "1") != "0") if segmented is None else bool(segmented)
        self.ts = TrainStep(self.net, self.fpn, self.head if yolox_loss else None, lr=2e-4, weight_decay=0.0, clip_value=1.0, world=world,
                            segmented=self.segmented, defer_dw=defer_dw, dw_rows=dw_rows, dw_discard=dw_discard, cuts=cuts)
        self.flat, self.opt = self.ts.flat, self.ts.opt
        # seq_len > 1 (opt-in, --seq-len): the reference's BPTT step shape (modules/detection.py:141-177): L timesteps with the
        # recurrent states carried, PAFPN + loss on the last one, one backward through time.  Default 1 = BASELINE's metric.
        self.xs = [synthetic_events(BATCH, HW, seed=rank + 1000 * t).to(dev) for t in range(seq_len)]
        if event_dtype == "uint8":      # the dataset's storage type (data/genx_utils/sequence_base.py:88-98): stays bytes up to the stem conv
            self.xs = [x.to(torch.uint8) for x in self.xs]
        self.x = self.xs[0]
        self.fwd_only = fwd_only or infer     # --fwd-only: backbone forward, the reference's own benchmark.py protocol (BASELINE config C2)
        self.loss = None
        self.P = None
        self.step0 = None
        self.graph = None
        self.use_graph = use_graph

    def fwd_only_pass(self):
        with torch.no_grad():
            feats, _states, P = self.net.forward_nhwc(self.x)
            if self.infer:
                pred = self.head.forward_nhwc(self.fpn.forward_nhwc(feats))
                self.loss = pred[..., 4].sum()
            else:
                self.loss = feats[4].sum()
        self.P, self.feats = P, feats

    def fwd_bwd(self):
        """forward + whole backward, no reduce / update (roofline leg, parity leg)"""
        if self.fwd_only:
            return self.fwd_only_pass()
        self.ts.forward(self.xs, None, self.labels, self.indices)
        for i in range(self.ts.n_segments()):
            self.ts.backward_segment(i)
        self.ts.flush_pending()       # deferred weight gradients: run what the segments parked, here on the same stream
        self.loss, self.P = self.ts.loss.detach(), self.ts.P

    def eager_step(self):
        if self.fwd_only:
            return self.fwd_only_pass()
        self.ts.step(self.xs, None, self.labels, self.indices)
        self.loss, self.P = self.ts.loss.detach(), self.ts.P

    def capture(self):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for it in range(2):
                if it == 0 and not self.fwd_only:
                    # loss, kept-token counts and every gradient AT THE INITIAL WEIGHTS: the cpu_baseline leg runs the oracle on
                    # the same weights and input and reports the disagreement in the JSON line (`parity`)
                    self.fwd_bwd()
                    self.step0 = {"loss": float(self.loss), "P": [int(p) for p in self.P], "grad": self.flat.grad.detach().cpu().clone()}
                n0 = self.fpn._sync_group.n_collectives if self.sync_bn else 0
                self.eager_step()
                if self.sync_bn:       # statistics all-reduces of ONE step (PAFPN + head share the group)
                    self.sync_collectives_per_step = self.fpn._sync_group.n_collectives - n0
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.loss_first = float(self.loss)
        if not self.use_graph:
            return False
        try:
            if self.fwd_only:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    self.fwd_only_pass()
                self.graph = g
            else:
                # thread_local: calls made by other threads (the RCCL watchdog of torch.distributed) must not invalidate the capture
                self.ts.capture(self.xs, None, self.labels, self.indices)
                self.graph = self.ts
                self.loss, self.P = self.ts.loss, self.ts.P
            return True
        except Exception as e:  # noqa: BLE001
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            self.graph = None
            torch.cuda.synchronize()
            if self.segmented and not self.fwd_only:
                # the segmented step (three graphs + side-stream bucket updates) could not be captured: fall back to the plain step
                # (one graph for forward + backward, all-reduce + AdamW behind it) rather than to eager launches
                try:
                    self.ts.segmented = self.segmented = False
                    self.ts.capture(self.xs, None, self.labels, self.indices)
                    self.graph = self.ts
                    self.loss, self.P = self.ts.loss, self.ts.P
                    print("[bench] captured the unsegmented step instead", file=sys.stderr)
                    return True
                except Exception as e2:  # noqa: BLE001
                    print(f"[bench] unsegmented capture failed too ({type(e2).__name__}: {e2})", file=sys.stderr)
                    self.graph = None
                    torch.cuda.synchronize()
            return False

    def step(self):
        if self.graph is not None:
            self.graph.replay()
        else:
            self.eager_step()


def parity_vs_oracle(tr, p, f, loss_cpu, P_cpu):
    """the GPU leg's first step (initial weights, rank 0's input) against the oracle's first step on the same weights and input:
    loss, kept-token counts, and the gradient of every parameter tensor (max-norm relative error; the two tensors behind the
    scoring ReLU are reported separately, see tests/test_gpu_parity.py)."""
    s0 = tr.step0
    names = {}
    for mod, pre in ((tr.net, "net."), (tr.fpn, "fpn.")):
        for k, v in mod.named_parameters():
            names.setdefault(id(v), pre + k)
    worst, worst_kink, off, n = (0.0, None), (0.0, None), 0, 0
    for prm, off in zip(tr.flat.params, tr.flat.offsets):
        k = prm.numel()
        name = names[id(prm)]
        from sast_amd.dist import _phys_view
        g = _phys_view(s0["grad"][off:off + k], prm)
        ref = (p if name.startswith("net.") else f)[name[4:]].grad
        if ref is None:
            continue
        err = float((g - ref).abs().max()) / (float(ref.abs().max()) + 1e-30)
        n += 1
        if "to_scores." in name:
            worst_kink = max(worst_kink, (err, name))
        else:
            worst = max(worst, (err, name))
    return {"loss_gpu": s0["loss"], "loss_cpu": loss_cpu, "loss_rel_err": abs(s0["loss"] - loss_cpu) / abs(loss_cpu),
            "kept_tokens_equal": s0["P"] == [int(v) for v in P_cpu], "grad_tensors_compared": n,
            "grad_max_rel_err": worst[0], "grad_max_rel_err_tensor": worst[1],
            "grad_max_rel_err_behind_scoring_relu": worst_kink[0], "note": "first step at the initial weights, max-norm relative error per tensor"}


def _physical_cores():
    """distinct (physical id, core id) pairs of /proc/cpuinfo (None where the file does not say)"""
    try:
        cores, phys = set(), None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                cores.add((phys, line.split(":")[1].strip()))
        return len(cores) or None
    except OSError:
        return None


def cpu_baseline(amp, init_state, seconds_budget=25.0, fwd_only=False, tr=None):
    """the oracle (torch-CPU port of the reference path) timed on this box's host cores on a bounded sample,
    on the same initial weights and the same input as rank 0's GPU leg."""
    from oracle import sast_oracle as O
    ocfg = O.BackboneCfg(in_res_hw=HW, partition_size=PART, amp=amp)
    p = {k: v.clone().float().requires_grad_(True) for k, v in init_state["net"].items()}
    f = {k: (v.clone().float().requires_grad_(True) if ("running" not in k and "num_batches" not in k) else v.clone())
         for k, v in init_state["fpn"].items()}
    x = synthetic_events(BATCH, HW, seed=0)

    def one():
        if fwd_only:
            with torch.no_grad():
                O.backbone(x, None, p, ocfg)
            return
        for t in list(p.values()) + list(f.values()):
            t.grad = None
        out, _s, P_cpu = O.backbone(x, None, p, ocfg)
        outs = O.pafpn(out, f, training=True)
        loss = O.proxy_loss(outs)
        loss.backward()
        return float(loss), P_cpu

    # MKL/OpenMP over-subscription makes "all hardware threads" the slowest choice on big hosts: probe a few thread
    # counts with one step each and time the sample at the fastest (reported in `cores`)
    first = one()
    parity = parity_vs_oracle(tr, p, f, *first) if (tr is not None and tr.step0 is not None and first is not None) else None
    best = (float("inf"), torch.get_num_threads())
    for nt in sorted({torch.get_num_threads(), 64, 32, 16, 8}):
        if nt > (os.cpu_count() or 1):
            continue
        torch.set_num_threads(nt)
        t0 = time.perf_counter()
        one()
        dt = time.perf_counter() - t0
        if dt < best[0]:
            best = (dt, nt)
    cores = best[1]
    torch.set_num_threads(cores)
    n, t0 = 0, time.perf_counter()
    while True:
        one()
        n += 1
        el = time.perf_counter() - t0
        if el > seconds_budget or n >= (32 if fwd_only else 8):
            break
    return parity, {"value": BATCH * n / el, "unit": "frames/s", "cores": cores, "threads": cores, "host_logical_cpus": os.cpu_count(),
            "host_physical_cores": _physical_cores(), "kind": "port",
            "sample": (f"{n} backbone forward passes of the same workload (B={BATCH}, {HW[0]}x{HW[1]}), " if fwd_only else
                       f"{n} fwd+bwd steps of the same workload (B={BATCH}, {HW[0]}x{HW[1]}, backbone+PAFPN, proxy loss), ") +
                      f"oracle/sast_oracle.py, torch {torch.__version__} CPU, {cores} threads (the fastest of 8 / 16 / 32 / 64 on this host), no optimizer step"}


def batch_scan(args, dev, world, ms_at_batch, batches=(1, 2)):
    """un-timed leg (round-5 verdict item 6): the same step at smaller batches -> least-squares line ms = a + b * B over B in {1, 2, BATCH}:
    `a` is the batch-INDEPENDENT part of the step (the latency floor of its chain of dependent launches), `b` the cost per sample"""
    global BATCH
    pts = {BATCH: ms_at_batch}
    keep = BATCH
    try:
        for b in batches:
            if b >= keep:
                continue
            BATCH = b
            tr = Trainer(dev, args.amp, world, use_graph=not args.no_graph, seq_len=args.seq_len, yolox_loss=args.loss == "yolox", segmented=args.segmented,
                         sync_bn=args.sync_bn, event_dtype=args.event_dtype, defer_dw=bool(args.defer_dw), dw_rows=(args.dw_min_rows, args.dw_max_rows),
                         cuts=tuple(int(c) for c in args.cuts.split(",")))
            tr.capture()
            for _ in range(10):
                tr.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(40):
                tr.step()
            torch.cuda.synchronize()
            pts[b] = 1e3 * (time.perf_counter() - t0) / 40
            del tr
            torch.cuda.empty_cache()
    finally:
        BATCH = keep
    xs, ys = list(pts.keys()), list(pts.values())
    n, sx, sy = len(xs), sum(xs), sum(ys)
    sxx, sxy = sum(x * x for x in xs), sum(x * y for x, y in zip(xs, ys))
    slope = (n * sxy - sx * sy) / (n * sxx - sx * sx)
    return {"ms_at_batch": {str(k): round(v, 4) for k, v in sorted(pts.items())}, "batch_independent_ms": (sy - slope * sx) / n, "ms_per_sample": slope}


def _fused_forward_in_use(amp):
    from sast_amd import functional as SF, _lib as SL
    from sast_amd.layers.sast import FUSED_FORWARD_MAX_AMP
    rows = BATCH * (HW[0] // 4) * (HW[1] // 4)
    T = PART[0] * PART[1]
    return bool(SF._FUSED_ENABLE and amp <= FUSED_FORWARD_MAX_AMP and rows >= SF._FUSED_MIN_ROWS and SL.lib().sast_mswsa_fused_ws_floats(64, 160, T, 32, 0) > 0)


def main():
    global BATCH, HW, PART
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300, help="timed steps (default = the reference's benchmark.py protocol: 300 timed iterations, benchmark.py:52-64)")
    ap.add_argument("--warmup", type=int, default=50, help="untimed warm-up steps (reference protocol: 50)")
    ap.add_argument("--amp", type=float, default=2e-4, help="attention_cfg.AMP (controls the kept-token fraction)")
    ap.add_argument("--batch", type=int, default=BATCH, help="samples per GPU (BASELINE config: 4; sparsity sweep C5: 8)")
    ap.add_argument("--res", choices=["1mpx", "gen1", "1mpx-split1"], default="1mpx", help="1mpx: 384x640 partition (6,10); gen1: 256x320 (8,10); "
                    "1mpx-split1: 384x640 with partition_split_32 1 (config/modifier.py:28-37) = partitions of 12x20 = 240 tokens")
    ap.add_argument("--seq-len", type=int, default=1, help="timesteps per step with recurrent state + BPTT (1 = BASELINE metric)")
    ap.add_argument("--fwd-only", action="store_true", help="backbone forward only (reference benchmark.py protocol; BASELINE config C2 with --res gen1)")
    ap.add_argument("--loss", choices=["proxy", "yolox"], default="proxy", help="proxy: sum mean(out^2) over the PAFPN outputs (BASELINE "
                    "metric); yolox: YOLOX head + SimOTA loss on synthetic boxes (the reference's real training objective)")
    ap.add_argument("--infer", action="store_true", help="backbone + PAFPN + YOLOX head, eval mode, forward only (decoded predictions)")
    ap.add_argument("--segmented", dest="segmented", action="store_true", default=None, help="backward in 3 segments with bucketed all-reduce + "
                    "AdamW on a side stream overlapping the remaining backward (default for N > 1; with N = 1 the same path, all-reduce a no-op)")
    ap.add_argument("--no-segmented", dest="segmented", action="store_false")
    ap.add_argument("--label-every", type=int, default=0, help="with --loss yolox and --seq-len L: labels on the samples b of timestep t with "
                    "(t + b) %% k == 0 (and on all of the last timestep); their features are gathered into one head call (label-sparse step)")
    ap.add_argument("--precision", choices=["f32", "bf16"], default="f32", help="f32: the product path (BASELINE metric).  bf16: the separately "
                    "built reduced-precision library (GEMM operands rounded to bf16, fp32 accumulate; the arithmetic class of the reference's AMP-16 "
                    "experiments) -- reported as a different metric, index decisions differ from the fp32 reference's")
    ap.add_argument("--sync-bn", action="store_true", help="N > 1: SyncBatchNorm in the PAFPN / head like the reference's DDP runs (train.py:167); "
                    "the statistics all-reduces are captured into the hipGraphs on RCCL, eager step on gloo.  Default: per-rank batch statistics")
    ap.add_argument("--event-dtype", choices=["int32", "uint8"], default="int32", help="int32: the reference's benchmark.py protocol (`.int()`); "
                    "uint8: the dataset's storage type -- the event tensor stays bytes up to the stem conv's loaders")
    ap.add_argument("--defer-dw", type=int, default=int(os.environ.get("SAST_DEFER_DW", "0")), help="1: weight gradients off the backward chain "
                    "(training.TrainStep(defer_dw=True)): the dW jobs are parked and run on the side stream beside the next segment's chain")
    ap.add_argument("--cuts", default="3", help="segment boundaries of the segmented backward: backbone stages (0-based) whose input is a cut, "
                    "e.g. 3,2,1 = one segment per stage (default 3: PAFPN | stage 4 | stages 3-1)")
    ap.add_argument("--dw-min-rows", type=int, default=0)
    ap.add_argument("--dw-max-rows", type=int, default=0, help="only weight-gradient jobs over at most this many rows are deferred (0 = all)")
    ap.add_argument("--dw-discard", action="store_true", help="TIMING PROBE: the parked jobs are dropped (the dX chain alone); gradients are wrong")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-batch-scan", action="store_true", help="skip the un-timed B = 1, 2 legs behind roofline.whole_step.batch_independent_ms")
    args = ap.parse_args()
    BATCH = args.batch
    if args.precision == "bf16":       # before anything imports sast_amd._lib; opt-in build: `python -m sast_amd.build --bf16`
        os.environ["SAST_LIB_PATH"] = os.path.join(ROOT, "sast_amd", "libsast_hip_bf16.so")
        if not os.path.exists(os.environ["SAST_LIB_PATH"]):
            raise SystemExit("--precision bf16 needs the opt-in library: python -m sast_amd.build --bf16")
    if args.res == "gen1":
        HW, PART = (256, 320), (8, 10)
    elif args.res == "1mpx-split1":
        PART = (12, 20)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus} (WORLD_SIZE={world})")
    ndev = torch.cuda.device_count()
    local = local % max(ndev, 1)          # (test rigs with fewer GPUs than ranks share a device; the driver gives one GPU per rank)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # SAST_SYNC_BN_FORCE=1 with --sync-bn and one rank: a ONE-rank RCCL group, so that the captured statistics all-reduces are real
    # RCCL calls (single-GPU check of the capture path; the collectives then move no data)
    one_rank_group = world == 1 and args.sync_bn and os.environ.get("SAST_SYNC_BN_FORCE", "0") == "1"
    if one_rank_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("SAST_DIST_BACKEND", "nccl")   # "nccl" is RCCL on ROCm; gloo only for plumbing tests
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        # proof that the collective backend really spans `world` ranks (not just that WORLD_SIZE says so): every rank contributes a one
        ones = torch.ones(1, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(ones)
        ranks_verified = int(round(float(ones.item())))
        _flush_c_stdio()
        if ranks_verified != world:
            raise SystemExit(f"all-reduce of ones returned {ranks_verified}, expected {world} ranks")
    else:
        ranks_verified = 1

    # Everything below runs on an explicit (non-default) HIP stream.  Measured on ROCm 7.2 / MI355X: a hipGraph launched on
    # the legacy NULL stream is NOT ordered after kernels enqueued on the NULL stream just before it, so with the optimizer
    # step outside the graph (N > 1) the next replay's gradient clear raced the AdamW kernel.  On a created stream the
    # replays and the eager all-reduce / AdamW are strictly ordered.
    run_stream = torch.cuda.Stream()
    run_stream.wait_stream(torch.cuda.current_stream())
    torch.cuda.set_stream(run_stream)
    tr = Trainer(dev, args.amp, world, use_graph=not args.no_graph, seq_len=args.seq_len, fwd_only=args.fwd_only, infer=args.infer,
                 yolox_loss=args.loss == "yolox", segmented=args.segmented, label_every=args.label_every, sync_bn=args.sync_bn,
                 event_dtype=args.event_dtype, defer_dw=bool(args.defer_dw), dw_rows=(args.dw_min_rows, args.dw_max_rows), dw_discard=args.dw_discard,
                 cuts=tuple(int(c) for c in args.cuts.split(",")))
    graphed = tr.capture()
    loss_first = tr.loss_first
    for _ in range(args.warmup):
        tr.step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tr.step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    # N > 1: how much of the gradient exchange is NOT hidden behind the backward -- a few extra, un-timed steps on every rank with
    # event pairs (main stream idle, side stream done); max over ranks
    exposed_ms, bucket_ms = None, None
    if (world > 1 or tr.segmented or args.defer_dw) and not tr.fwd_only:
        tr.ts.measure_exposed = True
        for _ in range(20):
            tr.step()
        exposed_ms = tr.ts.exposed_ms()
        bucket_ms = tr.ts.bucket_ms()
        tr.ts.measure_exposed = False
        if world > 1:
            t = torch.tensor([exposed_ms if exposed_ms is not None else -1.0], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            exposed_ms = float(t.item()) if float(t.item()) >= 0 else None

    if rank == 0:
        kept = [int(p) for p in tr.P]
        L = [(HW[0] // s) * (HW[1] // s) for s in (4, 8, 16, 32)]
        # BASELINE.json's metric string only for BASELINE's configuration (configs[2], and [3] for N > 1); any other run is labelled
        baseline_cfg = (args.res == "1mpx" and BATCH == 4 and args.seq_len == 1 and not (args.fwd_only or args.infer) and args.loss == "proxy"
                        and args.precision == "f32")
        metric = "frames/sec (B=4) SAST backbone fwd+bwd, 1Mpx 640x360" if baseline_cfg else (
            f"frames/sec (B={BATCH}) SAST " + ("backbone+PAFPN+head inference" if args.infer else "backbone fwd" if args.fwd_only else
                                               "backbone+PAFPN+YOLOX-loss fwd+bwd" if args.loss == "yolox" else "backbone fwd+bwd") +
            ({"1mpx": ", 1Mpx 640x360", "gen1": ", Gen1 304x240", "1mpx-split1": ", 1Mpx 640x360 partitions 12x20"}[args.res]) + (f", {args.seq_len} timesteps BPTT" if args.seq_len > 1 else "") +
            (", bf16 GEMM operands (reduced-precision library)" if args.precision == "bf16" else "") +
            " [not the BASELINE.json metric configuration]")
        res = {
            "metric": metric, "baseline_metric": baseline_cfg,
            "value": BATCH * args.seq_len * world * args.steps / el, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.precision == "f32" else "bf16 operands / f32 accumulate", "data": "synthetic",
            "config": {"workload": ({"1mpx": "1Mpx 640x360 (padded 384x640)", "gen1": "Gen1 304x240 (padded 256x320)",
                                    "1mpx-split1": "1Mpx 640x360 (padded 384x640), partition_split_32 1 (12x20 = 240-token partitions)"}[args.res]) +
                                   (" full SAST backbone + PAFPN + YOLOX head (3 classes), eval forward -> decoded predictions, " if args.infer else
                                    " full SAST backbone, forward only (benchmark.py protocol), " if args.fwd_only else
                                    " full SAST backbone + PAFPN + YOLOX head, SimOTA loss on 16 synthetic boxes/sample, fwd+bwd + AdamW, "
                                    if args.loss == "yolox" else " full SAST backbone + PAFPN, fwd+bwd + AdamW, ") +
                                   f"B={BATCH}/GPU, binary events (rand>{SPARSITY}), random-init, AMP={args.amp}",
                       "global_batch": BATCH * world, "seq_len": args.seq_len, "parallelism": f"dp{world}", "hipgraph": bool(graphed),
                       "gemm_arithmetic": {1: "fp32 products on the bf16 MFMA pipe: exact 3-way bf16 operand split, 6 MFMAs per product tile, fp32 accumulate",
                                           0: "v_mfma_f32_32x32x2_f32", 2: "operands rounded to bf16, one bf16 MFMA per tile step, fp32 accumulate (reduced precision)"}[
                                               __import__("sast_amd._lib", fromlist=["lib"]).lib().sast_mfma_split3()],
                       "library": os.path.relpath(__import__("sast_amd._lib", fromlist=["lib"]).loaded_path(), ROOT),
                       "product_library": __import__("sast_amd._lib", fromlist=["lib"]).is_product_library(),
                       "segmented_backward_overlap": bool(tr.segmented and not tr.fwd_only),
                       "deferred_weight_gradients": bool(args.defer_dw and not tr.fwd_only), "dw_rows_window": [args.dw_min_rows, args.dw_max_rows], "backward_cuts": args.cuts,
                       "dw_discard_TIMING_PROBE_gradients_invalid": bool(args.dw_discard),
                       # the dim-64 MS-WSA layers (stage 1) run their forward as ONE kernel (csrc/k_mswsa_fused.hip) when the rows and the
                       # block's AMP allow it (sast_amd/functional.py: _FUSED_MIN_ROWS, layers/sast.py: FUSED_FORWARD_MAX_AMP)
                       "fused_mswsa_forward": _fused_forward_in_use(args.amp),
                       "collective_ranks": world, "collective_ranks_verified": ranks_verified,
                       "collective_backend": (dist.get_backend() if dist.is_initialized() else None),
                       "gradient_bytes_per_rank": 4 * int(tr.flat.numel),
                       "allreduce_exposed_ms": exposed_ms,
                       # rank 0's all-reduce + AdamW time per gradient bucket (PAFPN[+head], stage 4, stages 3-1), on the stream that ran it
                       "allreduce_update_ms_per_bucket": ({str(k): round(v, 4) for k, v in bucket_ms.items()} if bucket_ms else None),
                       # the reference's DDP runs convert BatchNorm to SyncBatchNorm (train.py:167); false = per-rank batch statistics
                       "sync_batchnorm": bool(tr.sync_bn),
                       "sync_batchnorm_collectives_per_step": (tr.sync_collectives_per_step if tr.sync_bn else None),
                       "event_dtype": args.event_dtype,
                       "kept_token_fraction_per_stage": [round(k / (2 * l), 4) for k, l in zip(kept, L)],
                       "loss_first_step": loss_first, "loss": float(tr.loss),
                       "grads_finite": bool(torch.isfinite(tr.flat.grad).all()), "grad_absmax": float(tr.flat.grad.abs().max())},
        }
        # (the roofline leg re-runs the step on rank 0 ALONE: with SyncBatchNorm over several ranks its statistics all-reduces would wait
        # for ranks that are not there -- skipped, `roofline` is then absent from the line)
        if not args.no_roofline and not (tr.sync_bn and world > 1):
            from sast_amd.profiling import dominant_kernel_roofline
            res["roofline"] = dominant_kernel_roofline(tr, ms_per_step=res["ms_per_step"], hw=HW, batch=BATCH, seq_len=max(args.seq_len, 1),
                                                       pmc_applies=baseline_cfg and args.event_dtype == "int32" and not tr.sync_bn)
            ws = res["roofline"].get("whole_step")
            if ws is not None:
                ws["peak_memory_bytes"] = {"allocated": int(torch.cuda.max_memory_allocated()), "reserved": int(torch.cuda.max_memory_reserved())}
                if world == 1 and not args.no_batch_scan and not (args.fwd_only or args.infer) and BATCH > 2:
                    ws.update(batch_scan(args, dev, world, res["ms_per_step"]))
        if not args.no_cpu_baseline and world == 1 and args.seq_len == 1 and not args.infer and args.loss == "proxy" and args.precision == "f32":
            parity, res["cpu_baseline"] = cpu_baseline(args.amp, tr.init_state, fwd_only=args.fwd_only, tr=tr)
            if parity is not None:
                res["parity"] = parity
    if world > 1:
        dist.barrier()       # rank 0 may still be in its (rank-local) roofline leg: leave the group together
    _flush_c_stdio()         # RCCL's version banner sits in the C library's stdout buffer: out with it BEFORE the one JSON line
    if rank == 0:
        print(json.dumps(res), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()
        if world > 1:
            # the result is out and the group is gone: leave without the interpreter's teardown.  With 8 ranks the communicator /
            # socket destructors that run at exit abort now and then when the peers close together (seen with gloo on the CPU
            # tests: SIGABRT after every rank had finished) -- torch.distributed.run would report that as a failed run
            _flush_c_stdio()
            sys.stderr.flush()
            os._exit(0)


if __name__ == "__main__":
    main()
