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
    """content hash of the kernel sources: stamped into the PMC summary by tools/rocpd_pmc.py, compared by bench.py so that a
    traffic figure measured on older kernels is flagged instead of silently reported"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".cuh", ".h")):
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


def gemm_report(run_steps, n_steps: int = 3, per_shape: bool = False):
    """run_steps(n): executes n eager steps.  Returns [(name, calls, total_ms, total_flops)] sorted by time; per_shape: one row
    per (instantiation, problem shape), the shape appended to the name after " |"."""
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
        tag, n, ms, fl = line.rsplit("\t", 3)
        tag, _, shape = tag.partition(" |")
        rows.append((_short(tag) + (" |" + shape if shape else ""), int(n), float(ms), float(fl)))
    rows.sort(key=lambda r: -r[2])
    return rows


def dominant_kernel_roofline(trainer, n_steps: int = 3):
    # rank-local on purpose: this leg runs on rank 0 only, after the timed region -- it must not enter a collective (the
    # other ranks are already past it), so it replays forward + backward without the gradient all-reduce / optimizer step
    def run(n):
        for _ in range(n):
            trainer.fwd_bwd()

    rows = [r for r in gemm_report(run, n_steps) if not r[0].startswith("op:")]
    # GEMM launches are timed with hipExtLaunchKernelGGL start/stop events (stamped at the kernel's own begin / end, the
    # same quantity rocprofv3 reports); a plain record-launch-record bracket would add ~8 us (reported for reference)
    calib_ms = float(L.lib().sast_prof_calibrate(C.c_void_p(torch.cuda.current_stream().cuda_stream), 200))
    rows.sort(key=lambda r: -r[2])
    name, calls, ms, flops = rows[0]
    achieved = flops / (ms * 1e-3) / 1e12
    total_ms = sum(r[2] for r in rows)
    total_fl = sum(r[3] for r in rows)
    stamp = pmc_summary_stamp()
    split3 = L.lib().sast_mfma_split3() == 1
    pipe = {}
    if split3:
        # the default build executes an fp32 product tile as 6 bf16 MFMAs on an exact 3-way operand split: the contract's `peak`
        # stays the dense MFMA peak of the dtype the path computes in (f32: 157.3), the bound of the pipe actually used is given too
        eff = PEAK_BF16_MFMA_TFLOPS / SPLIT3_PRODUCTS
        pipe = {"executed_as": "fp32 operands split exactly into 3 bf16 terms, 6 v_mfma_f32_32x32x16_bf16 per product tile, fp32 accumulate "
                               "(error <= 2^-23 |x||y| per product); SAST_MFMA_SPLIT3=0 builds the v_mfma_f32_32x32x2_f32 form",
                "peak_of_executed_pipe": eff, "frac_of_executed_pipe": achieved / eff}
    return {"bound": "mfma", "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", **pipe,
            "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": pmc_traffic_bytes(name), "traffic_unit": "bytes/launch (rocprofv3 PMC, "
            "profiles/pmc_hbm_traffic_latest.json)", "traffic_profile_csrc_sha": stamp, "traffic_stale": stamp != csrc_sha(), "kernel": name,
            "launches_per_step": calls / n_steps, "avg_launch_us": 1e3 * ms / calls,
            "algorithmic_gflop_per_launch": flops / calls / 1e9,
            "all_gemm_kernels": {"ms_per_step": total_ms / n_steps, "gflop_per_step": total_fl / n_steps / 1e9,
                                 "achieved_tflops": total_fl / (total_ms * 1e-3) / 1e12},
            "plain_event_bracket_overhead_us": 1e3 * calib_ms,
            "method": "hipExtLaunchKernelGGL start/stop events on the launch stream for every launch of the GEMM-template kernels "
                      "(libsast_hip sast_prof_*); eager, un-timed extra steps"}
