"""Roofline leg of bench.py: HIP-event timing (on the launch stream, inside libsast_hip.so) of every launch of the
GEMM-template kernels during a few extra, un-timed eager steps; reports the dominant instantiation.

Peak: 157.3 TFLOP/s fp32 matrix (v_mfma_f32_32x32x2_f32, /opt/skills/guides/MI355X_MICROARCH.md) -- all GEMM-shaped work
of this path is exact fp32 because index-exact token selection forbids reduced precision upstream of a selection."""
from __future__ import annotations

import ctypes as C
import re

import torch

import json
import os

from . import _lib as L

PEAK_F32_MFMA_TFLOPS = 157.3          # 256 CUs x 4 SIMDs x 64 flop/cycle x 2.4 GHz (v_mfma_f32_32x32x2_f32)
PEAK_BF16_MFMA_TFLOPS = 2516.6        # 256 CUs x 4 SIMDs x 1024 flop/cycle x 2.4 GHz (v_mfma_f32_32x32x16_bf16), dense
SPLIT3_PRODUCTS = 6                   # bf16 MFMAs per fp32 product tile in the default build (gemm.cuh: split3)
_PMC_SUMMARY = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_hbm_traffic_latest.json")


_TILE_DEFAULTS = (None, None, None, None, None, "16", "1", "2", "0")   # Tile<BM, BN, WM, WN, G, BK=16, KS=1, PF=2, OCC=0>


def _norm_kernel(name: str) -> str:
    """canonical spelling of a GEMM-template instantiation: rocprofv3 prints every Tile<> argument and the split flag as
    true/false, __PRETTY_FUNCTION__ (the roofline leg) omits defaulted arguments and we tag split launches ourselves."""
    name = re.sub(r"\(.*$", "", name).replace("> >", ">>")                # drop the argument list rocprof appends

    def fill(m):
        args = [a.strip() for a in m.group(1).split(",")]
        return "Tile<" + ", ".join(args + list(_TILE_DEFAULTS[len(args):])) + ">"

    name = re.sub(r"Tile<([^>]*)>", fill, name)
    name = name.replace(", split>", ", true>")
    if name.startswith("gemm_kernel<") and not re.search(r", (true|false)>$", name):
        name = name[:-1] + ", false>"
    return re.sub(r"\s+", " ", name)


def csrc_sha() -> str:
    """content hash of the PRODUCT kernel sources (what libsast_hip.so is built from: build.SOURCES + the headers; the tools-only
    translation units k_test.hip / k_dma_test.hip do not count): stamped into the PMC summary by tools/rocpd_pmc.py, compared by bench.py
    so that a traffic figure measured on older kernels is flagged instead of silently reported"""
    import hashlib
    from .build import SOURCES
    h = hashlib.sha256()
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    for fn in sorted(os.listdir(d)):
        if fn in SOURCES or fn.endswith((".cuh", ".h")):
            with open(os.path.join(d, fn), "rb") as f:
                h.update(fn.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def pmc_summary_stamp():
    try:
        with open(_PMC_SUMMARY) as f:
            return json.load(f).get("csrc_sha")
    except (OSError, ValueError):
        return None


def pmc_traffic_bytes(kernel_name: str):
    """HBM bytes per launch of `kernel_name` from the committed rocprofv3 --pmc summary (tools/rocpd_pmc.py: two separate
    passes FETCH_SIZE / WRITE_SIZE, FETCH doubled per the gfx950 note of MI355X_MICROARCH.md); None when not profiled."""
    try:
        with open(_PMC_SUMMARY) as f:
            ks = json.load(f)["kernels"]
    except (OSError, ValueError, KeyError):
        return None
    want = _norm_kernel(kernel_name)
    for k, ent in ks.items():
        if _norm_kernel(k) == want:
            return float(ent["hbm_bytes_per_launch"])
    return None


def _clean(t: str) -> str:
    return re.sub(r"sast::|\(anonymous namespace\)::", "", t).strip()


def _short(tag: str) -> str:
    """kernel name (as rocprofv3 spells it, minus defaulted Tile<> arguments) from the __PRETTY_FUNCTION__ of its launcher"""
    m = re.search(r"launch_gemm_dual\b.*\[TS = (.*?), LA1 = (.*?), LB1 = (.*?), EP1 = (.*?), TP = (.*?), LA2 = (.*?), LB2 = (.*?), EP2 = (.*?)\]", tag)
    if m:
        g = [_clean(m.group(i)) for i in range(1, 9)]
        return (f"gemm_dual_kernel<GemmJob<{g[0]}, {g[1]}, {g[2]}, {g[3]}, true>, GemmJob<{g[4]}, {g[5]}, {g[6]}, {g[7]}, false>>")
    m = re.search(r"launch_gemm(_split)?\b.*\[T = (.*?), LA = (.*?), LB = (.*?), EP = (.*?)\]", tag)
    if not m:
        return tag[:120]
    names = [_clean(m.group(i)) for i in (2, 3, 4, 5)]
    return f"gemm_kernel<{names[0]}, {names[1]}, {names[2]}, {names[3]}{', split' if m.group(1) else ''}>"


def kernel_report(run_steps, n_steps: int = 3, per_shape: bool = False):
    """run_steps(n): executes n eager steps with every launch of the GEMM-template kernels AND of the attention kernels bracketed by
    kernel-exact HIP events.  Returns rows {name, calls, ms, flops, bytes} sorted by time (op-level scopes are named "op:...");
    per_shape: one row per (instantiation, problem shape), the shape appended to the name after " |"."""
    lib = L.lib()
    torch.cuda.synchronize()
    lib.sast_prof_enable(2 if per_shape else 1)
    try:
        run_steps(n_steps)
        torch.cuda.synchronize()
        need = lib.sast_prof_report(None, 0)
        buf = C.create_string_buffer(int(need) + 16)
        lib.sast_prof_report(buf, len(buf))
    finally:
        lib.sast_prof_enable(0)
    rows = []
    for line in buf.value.decode().splitlines():
        tag, n, ms, fl, by = line.rsplit("\t", 4)
        tag, _, shape = tag.partition(" |")
        name = tag if tag.startswith(("attn_", "op:")) else _short(tag)
        rows.append({"name": name + (" |" + shape if shape else ""), "calls": int(n), "ms": float(ms), "flops": float(fl), "bytes": float(by)})
    rows.sort(key=lambda r: -r["ms"])
    return rows


def gemm_report(run_steps, n_steps: int = 3, per_shape: bool = False):
    """(tools) the same as tuples (name, calls, total_ms, total_flops)"""
    return [(r["name"], r["calls"], r["ms"], r["flops"]) for r in kernel_report(run_steps, n_steps, per_shape)]


PEAK_HBM_TBS = 8.0                    # HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md (measured copy rate: 6.29 TB/s)


def algorithmic_bytes_per_step(hw, batch, embed_dim=64, in_bytes=4, fpn_channels=(128, 256, 512), fpn_depth=0.67, n_params=0,
                               fwd_only=False, with_fpn=True):
    """SURVEY.md section 8(d) rule -- every LOGICAL operator reads its input once and writes its output once, fp32 activations,
    weights once per batch: per stage 12 A_s + 4 L_s with A_s = 4 L_s C_s bytes per frame (downsample write A; STP read A + write A +
    token scores; each of the two MS-WSA layers read A + write A; ConvLSTM read 3A + write 2A), the event tensor once
    (20 Hp Wp in_bytes), the PAFPN's 32 conv-BN-SiLU units (in + out), the weights (4 bytes per parameter) once.  The backward is
    counted as 2x the forward (saved activations re-read, gradients written), the optimizer as 28 bytes per parameter (p, g, m, v read;
    p, m, v written).  Dense (every token kept): an upper bound of the algorithmic traffic of a sparse step."""
    H, W = hw
    fwd = 20.0 * H * W * in_bytes
    for s in range(4):
        L_s, C_s = (H >> (2 + s)) * (W >> (2 + s)), embed_dim << s
        fwd += 12.0 * 4 * L_s * C_s + 4.0 * L_s
    if with_fpn:
        c0, c1, c2 = fpn_channels
        p0, p1, p2 = ((H >> k) * (W >> k) for k in (3, 4, 5))
        n = round(3 * fpn_depth)

        def csp(cin, cout, px):
            h = cout // 2
            return px * (2 * (cin + h) + n * ((h + h) + (h + h)) + (2 * h + cout))

        units = p2 * (c2 + c1) + csp(2 * c1, c1, p1) + p1 * (c1 + c0) + csp(2 * c0, c0, p0) + (p0 * c0 + p1 * c0) + csp(2 * c0, c1, p1) + \
            (p1 * c1 + p2 * c1) + csp(2 * c1, c2, p2)
        fwd += 4.0 * units
    fwd *= batch
    fwd += 4.0 * n_params
    return fwd if fwd_only else 3.0 * fwd + 28.0 * n_params


def pmc_bytes_per_step():
    """whole-step HBM bytes of the committed rocprofv3 --pmc summary (None when the summary does not say how many steps it covers)"""
    try:
        with open(_PMC_SUMMARY) as f:
            d = json.load(f)
        return float(d["total_bytes"]) / float(d["steps"]) if d.get("steps") else None
    except (OSError, ValueError, KeyError):
        return None


_KT_SUMMARY = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "kernel_trace_latest.json")


def kernel_trace_per_step():
    """(kernel ms per step, dispatches per step, csrc stamp) of the committed rocprofv3 --kernel-trace summary of the headline step
    (tools/rocpd_stats.py --json), or (None, None, None)"""
    try:
        with open(_KT_SUMMARY) as f:
            d = json.load(f)
        return float(d["kernel_ms_per_step"]), float(d["dispatches_per_step"]), d.get("csrc_sha")
    except (OSError, ValueError, KeyError):
        return None, None, None


def library_launches_per_step(trainer, n: int = 2):
    """kernel launches libsast_hip enqueues for ONE eager forward + backward (include/sast_hip.h: sast_launch_count; the optimizer's
    launches and ATen's -- gradient clear, the loss's ones -- are not in it)"""
    lib = L.lib()
    trainer.fwd_bwd()
    c0 = int(lib.sast_launch_count())
    for _ in range(n):
        trainer.fwd_bwd()
    torch.cuda.synchronize()
    return (int(lib.sast_launch_count()) - c0) / n


def dominant_kernel_roofline(trainer, n_steps: int = 3, ms_per_step=None, hw=None, batch=None, seq_len: int = 1, pmc_applies: bool = True):
    """pmc_applies: the committed rocprofv3 --pmc summary (profiles/pmc_hbm_traffic_latest.json) was taken on THIS configuration (bench.py:
    the BASELINE configuration); otherwise `traffic` / `bytes_counter` are null -- counter bytes of another shape say nothing here"""
    # rank-local on purpose: this leg runs on rank 0 only, after the timed region -- it must not enter a collective (the
    # other ranks are already past it), so it replays forward + backward without the gradient all-reduce / optimizer step
    def run(n):
        for _ in range(n):
            trainer.fwd_bwd()

    rows = [r for r in kernel_report(run, n_steps) if not r["name"].startswith("op:")]
    # launches are timed with hipExtLaunchKernelGGL start/stop events (stamped at the kernel's own begin / end, the same
    # quantity rocprofv3 reports); a plain record-launch-record bracket would add ~8 us (reported for reference)
    calib_ms = float(L.lib().sast_prof_calibrate(C.c_void_p(torch.cuda.current_stream().cuda_stream), 200))
    top = rows[0]                                                # the kernel with the largest share of the step, GEMM template or attention
    name, calls, ms, flops, nbytes = top["name"], top["calls"], top["ms"], top["flops"], top["bytes"]
    gemm_rows = [r for r in rows if not r["name"].startswith("attn_")]
    attn_rows = [r for r in rows if r["name"].startswith("attn_")]
    achieved = flops / (ms * 1e-3) / 1e12
    achieved_tbs = nbytes / (ms * 1e-3) / 1e12
    stamp = pmc_summary_stamp()
    split3 = L.lib().sast_mfma_split3() == 1
    is_gemm = not name.startswith("attn_")
    pipe = {}
    if split3 and is_gemm:
        # the default build executes an fp32 product tile as 6 bf16 MFMAs on an exact 3-way operand split: the contract's `peak`
        # stays the dense MFMA peak of the dtype the path computes in (f32: 157.3), the bound of the pipe actually used is given too
        eff = PEAK_BF16_MFMA_TFLOPS / SPLIT3_PRODUCTS
        pipe = {"executed_as": "fp32 operands split exactly into 3 bf16 terms, 6 v_mfma_f32_32x32x16_bf16 per product tile, fp32 accumulate "
                               "(error <= 2^-23 |x||y| per product); SAST_MFMA_SPLIT3=0 builds the v_mfma_f32_32x32x2_f32 form",
                "peak_of_executed_pipe": eff, "frac_of_executed_pipe": achieved / eff}
    elif split3:
        # k_attn_mfma.hip / k_mswsa_fused.hip run the same bf16x3 split: six v_mfma_f32_32x32x16_bf16 per 32x32x16 product tile
        eff = PEAK_BF16_MFMA_TFLOPS / SPLIT3_PRODUCTS
        pipe = {"executed_as": "fp32 operands split exactly into 3 bf16 terms, 6 v_mfma_f32_32x32x16_bf16 per product tile, fp32 accumulate; "
                               "softmax in the MFMA C layout",
                "peak_of_executed_pipe": eff, "frac_of_executed_pipe": achieved / eff}
    # which roof is nearer for this kernel: time at the MFMA peak vs time at the HBM peak for its algorithmic work
    t_mfma, t_hbm = flops / (PEAK_F32_MFMA_TFLOPS * 1e12), nbytes / (PEAK_HBM_TBS * 1e12)
    bound = "mfma" if t_mfma >= t_hbm else "hbm"
    out = {"bound": bound,
           "achieved": achieved if bound == "mfma" else 1e3 * achieved_tbs,
           "peak": PEAK_F32_MFMA_TFLOPS if bound == "mfma" else 1e3 * PEAK_HBM_TBS,
           "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
           "frac": (achieved / PEAK_F32_MFMA_TFLOPS) if bound == "mfma" else achieved_tbs / PEAK_HBM_TBS,
           **pipe,
           "achieved_tflops": achieved, "frac_of_mfma_peak": achieved / PEAK_F32_MFMA_TFLOPS,
           "achieved_algorithmic_gbs": 1e3 * achieved_tbs, "frac_of_hbm_peak": achieved_tbs / PEAK_HBM_TBS,
           "traffic": pmc_traffic_bytes(name) if pmc_applies else None,
           "traffic_unit": "bytes/launch (rocprofv3 PMC, profiles/pmc_hbm_traffic_latest.json; null when that summary is of another configuration)",
           "traffic_over_algorithmic": (pmc_traffic_bytes(name) / (nbytes / calls)) if (pmc_applies and pmc_traffic_bytes(name) and nbytes) else None,
           "traffic_profile_csrc_sha": stamp, "traffic_stale": stamp != csrc_sha(), "kernel": name,
           "launches_per_step": calls / n_steps, "avg_launch_us": 1e3 * ms / calls,
           "algorithmic_gflop_per_launch": flops / calls / 1e9, "algorithmic_mbytes_per_launch": nbytes / calls / 1e6,
           "ranked_over": "every launch of the GEMM-template kernels and of the attention kernels (kernel-exact HIP events)",
           "all_gemm_kernels": {"ms_per_step": sum(r["ms"] for r in gemm_rows) / n_steps, "gflop_per_step": sum(r["flops"] for r in gemm_rows) / n_steps / 1e9,
                                "achieved_tflops": sum(r["flops"] for r in gemm_rows) / (sum(r["ms"] for r in gemm_rows) * 1e-3 + 1e-30) / 1e12},
           "all_attention_kernels": {"ms_per_step": sum(r["ms"] for r in attn_rows) / n_steps, "gflop_per_step": sum(r["flops"] for r in attn_rows) / n_steps / 1e9,
                                     "achieved_tflops": sum(r["flops"] for r in attn_rows) / (sum(r["ms"] for r in attn_rows) * 1e-3 + 1e-30) / 1e12,
                                     "achieved_algorithmic_gbs": sum(r["bytes"] for r in attn_rows) / (sum(r["ms"] for r in attn_rows) * 1e-3 + 1e-30) / 1e9},
           "plain_event_bracket_overhead_us": 1e3 * calib_ms,
           "method": "hipExtLaunchKernelGGL start/stop events on the launch stream for every launch of the GEMM-template and attention "
                     "kernels (libsast_hip sast_prof_*); eager, un-timed extra steps; algorithmic FLOPs / bytes from the device-side row and "
                     "kept-token counts"}
    if ms_per_step is not None and hw is not None:
        # the whole step against its roofs: algorithmic FLOPs of all matrix work (GEMMs + attention) at the f32-MFMA peak, algorithmic
        # bytes (SURVEY 8d rule, dense upper bound) at the HBM peak; T_roof = the larger of the two; frac = T_roof / measured step time
        n_params = int(trainer.flat.numel) if getattr(trainer, "flat", None) is not None else 0
        gflop = sum(r["flops"] for r in rows) / n_steps / 1e9
        fwd_only = bool(getattr(trainer, "fwd_only", False))
        b_alg = algorithmic_bytes_per_step(hw, batch, n_params=n_params, fwd_only=fwd_only, with_fpn=not fwd_only or bool(getattr(trainer, "infer", False)))
        if seq_len > 1:      # a BPTT step runs the backbone on every timestep (the measured gflop cover all of them); weights / optimizer once
            once = (4.0 if fwd_only else 3.0 * 4.0 + 28.0) * n_params
            b_alg = (b_alg - once) * seq_len + once
        t_m, t_h = gflop / PEAK_F32_MFMA_TFLOPS, b_alg / (PEAK_HBM_TBS * 1e12) * 1e3      # ms
        b_cnt = pmc_bytes_per_step() if pmc_applies else None
        kt_ms, kt_n, kt_sha = kernel_trace_per_step() if pmc_applies else (None, None, None)
        out["whole_step"] = {"gflop": gflop, "bytes_algorithmic": b_alg, "bytes_counter": b_cnt,
                             "bytes_counter_over_algorithmic": (b_cnt / b_alg) if b_cnt else None,
                             "t_mfma_ms": t_m, "t_hbm_ms": t_h, "t_roof_ms": max(t_m, t_h), "bound": "mfma" if t_m >= t_h else "hbm",
                             "ms_per_step": ms_per_step, "frac": max(t_m, t_h) / ms_per_step,
                             # against the pipe the default build EXECUTES the products on (6 bf16 MFMAs per fp32 product tile: 2516.6 / 6 =
                             # 419.4 TFLOP/s of fp32-equivalent work) -- the stricter of the two fractions
                             "t_executed_pipe_ms": (gflop / (PEAK_BF16_MFMA_TFLOPS / SPLIT3_PRODUCTS)) if split3 else t_m,
                             "frac_of_executed_pipe": (max(gflop / (PEAK_BF16_MFMA_TFLOPS / SPLIT3_PRODUCTS), t_h) if split3 else max(t_m, t_h)) / ms_per_step,
                             # the launch structure (round-5 verdict item 6): the step is a chain of dependent launches at their latency floor
                             "kernel_ms_per_step": kt_ms, "dispatches_per_step": kt_n, "kernel_trace_csrc_sha": kt_sha,
                             "kernel_trace_stale": (kt_sha != csrc_sha()) if kt_sha else None,
                             "library_launches_fwd_bwd": library_launches_per_step(trainer),
                             "achieved_tflops": gflop / ms_per_step, "achieved_counter_tbs": (b_cnt / (ms_per_step * 1e-3) / 1e12) if b_cnt else None,
                             "note": "gflop: algorithmic 2*M*N*K of every GEMM / conv + 4*C*sum K_m^2 (x2.5 backward) of the attention launches, from "
                                     "device-side counts; bytes_algorithmic: SURVEY 8d rule (dense upper bound, backward = 2x forward, optimizer 28 B/param); "
                                     "bytes_counter: rocprofv3 PMC (2*FETCH_SIZE + WRITE_SIZE) per step of the HEADLINE configuration, valid when "
                                     "traffic_stale is false; ms_per_step: one whole step (all timesteps of a sequence); kernel_ms_per_step / "
                                     "dispatches_per_step: sum of kernel durations and dispatch count of one replayed step in the committed rocprofv3 "
                                     "kernel trace (profiles/kernel_trace_latest.json, HEADLINE configuration); library_launches_fwd_bwd: launches "
                                     "this library enqueued for one eager forward + backward, counted live"}
    return out
