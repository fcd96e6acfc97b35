"""ctypes binding of libsast_hip.so (include/sast_hip.h).  Fails loudly when the library is missing:
there is NO CPU / PyTorch fallback for the hot path."""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB_PATH = os.path.join(_HERE, "libsast_hip.so")
# SAST_LIB_PATH: an A/B or variant build instead of the in-tree product library (tools / measurement scripts only).  The override is
# announced on stderr when the library is loaded and `loaded_path()` reports it (bench.py records it in its JSON line), so a stale
# variable in a shell cannot silently put a parity or benchmark claim on another binary.
LIB_PATH = os.environ.get("SAST_LIB_PATH") or DEFAULT_LIB_PATH
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "sast_hip.h")

P = C.c_void_p
I32 = C.c_int32
F32 = C.c_float

DT_F32, DT_I32, DT_U8 = 0, 1, 2


def _struct(name, spec):
    """spec: list of (ctype, 'a b c') -> ctypes.Structure with fields in order."""
    fields = []
    for ct, names in spec:
        for n in names.split():
            fields.append((n, ct))
    return type(name, (C.Structure,), {"_fields_": fields})


SastDownArgs = _struct("SastDownArgs", [
    (I32, "B H W Cin Cout factor"),
    (P, "x w ln_w ln_b pe conv_out mean rstd y dy dx dw d_ln_w d_ln_b ws"),
    (I32, "x_dtype no_overlap"),
])
SastScoreArgs = _struct("SastScoreArgs", [
    (I32, "B L C r_stride"), (F32, "amp"),
    (P, "xp r ws_w ws_b wc scale s xw tok dxw dxp d_ws_w d_ws_b d_wc ws dscale_ws"),
])
SastSel = _struct("SastSel", [(P, "win_keep mask K row_off win_rank counts tok_slot row_tok pack_rows row_seg")])
SastMswsaArgs = _struct("SastMswsaArgs", [
    (I32, "B H W C ph pw mode inner"), (F32, "eps"), (I32, "cb_tps dim_head mlp_act"),
    (P, "xin out"), (SastSel, "sel"),
    (P, "ln1_w ln1_b ln2_w ln2_b qkv_w qkv_b proj_w proj_b ls1 fc1_w fc1_b fc2_w fc2_b ls2"),
    (P, "mean1 rstd1 mean2 rstd2 S QKV O lse Y UG Hh"),
    (P, "dout dxin"),
    (P, "d_ln1_w d_ln1_b d_ln2_w d_ln2_b d_qkv_w d_qkv_b d_proj_w d_proj_b d_ls1 d_fc1_w d_fc1_b d_fc2_w d_fc2_b d_ls2"),
    (P, "ws cb_m cb_sum raw_ws drop1 drop2 drop_ws drop_mlp fused_ws act_w d_act_w"),
])
SastConvBn2Args = _struct("SastConvBn2Args", [
    (I32, "B H W Cin Cout ldx Cin1 ldx2 bn_ws_zeroed bn_red_done0 bn_red_done1 training ksize"), (F32, "momentum0 momentum1 eps0 eps1"),
    (P, "x x2 w0 w1 bn_w0 bn_w1 bn_b0 bn_b1 run_mean0 run_mean1 run_var0 run_var1 conv_out0 conv_out1 stats0 stats1 y0 y1 "
        "bn_ws0 bn_ws1 dy0 dy1 dw0 dw1 d_bn_w0 d_bn_w1 d_bn_b0 d_bn_b1 ws0 ws1 dx dx2 "
        "p_conv_out p_stats p_bn_w p_bn_b p_bn_ws p2_conv_out p2_stats p2_bn_w p2_bn_b p2_bn_ws"),
])
SastHeadGeom = _struct("SastHeadGeom", [(I32, "n_levels"), (I32 * 4, "H"), (I32 * 4, "W"), (F32 * 4, "stride")])
SastLstmArgs = _struct("SastLstmArgs", [
    (I32, "B L C"),
    (P, "x h0 c0 w b h1 c1 gates dh1 dc1 dx dh0 dc0 dw db ws dh1b drop"),
])
SastConvBnArgs = _struct("SastConvBnArgs", [
    (I32, "B H W Cin Cout ksize stride training ldx ldy lddy lddx bn_ws_zeroed bn_red_done Cin1 ldx2"), (F32, "momentum eps"),
    (P, "x w bn_w bn_b run_mean run_var conv_out stats y dy dx dw d_bn_w d_bn_b bn_ws ws x2 dx2 "
        "p_conv_out p_stats p_bn_w p_bn_b p_bn_ws p2_conv_out p2_stats p2_bn_w p2_bn_b p2_bn_ws dy2"),
    (I32, "sync_phase m_total groups"),
])

SastSampleGather = _struct("SastSampleGather", [
    (I32, "n_src n_out B _pad"), (C.c_size_t, "sample_floats"), (P * 32, "src"), (P * 32, "dsrc"), (P, "out"),
    (C.c_uint8 * 256, "t_of"), (C.c_uint8 * 256, "b_of"),
])
SastSampleMask = _struct("SastSampleMask", [(C.c_uint8 * 256, "sel")])

_SIGNATURES = {
    "sast_version": (C.c_int, []),
    "sast_mfma_split3": (C.c_int, []),
    "sast_nzratio": (C.c_int, [P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P, P, P]),
    "sast_nchw_to_nhwc": (C.c_int, [P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P, P]),
    "sast_nzratio_padded": (C.c_int, [P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P, P, P]),
    "sast_nchw_to_nhwc_padded": (C.c_int, [P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P, P]),
    "sast_nhwc_to_nchw": (C.c_int, [P, C.c_int, C.c_int, C.c_int, C.c_int, P, P]),
    "sast_input_prep": (C.c_int, [P] + [C.c_int] * 7 + [P, P, P, P]),
    "sast_input_prep_u8": (C.c_int, [P] + [C.c_int] * 6 + [P, P, P, P]),
    "sast_add_rows": (C.c_int, [P, P, P, C.c_int, C.c_int, C.c_int, P]),
    "sast_mean_square_fwd": (C.c_int, [P, P, C.c_int, P, P]),
    "sast_mean_square_bwd": (C.c_int, [P, P, C.c_int, P, C.c_int, P, P]),
    "sast_downsample_ln_fwd": (C.c_int, [C.POINTER(SastDownArgs), P]),
    "sast_downsample_ln_bwd": (C.c_int, [C.POINTER(SastDownArgs), P]),
    "sast_score_stp_fwd": (C.c_int, [C.POINTER(SastScoreArgs), P]),
    "sast_score_stp_bwd": (C.c_int, [C.POINTER(SastScoreArgs), P]),
    "sast_select": (C.c_int, [P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(SastSel), P]),
    "sast_select_packs": (C.c_int, [C.POINTER(SastSel), C.c_int, C.c_int, P]),
    "sast_head_pred_decode": (C.c_int, [P] * 9 + [C.c_int] * 5 + [C.c_float, C.c_int, C.c_int, C.c_int, P]),
    "sast_head_pred_fwd": (C.c_int, [P] * 10 + [C.c_int] * 5 + [C.c_float, C.c_int, C.c_int, C.c_int, P]),
    "sast_head_pred_bwd": (C.c_int, [P] * 14 + [C.c_int] * 7 + [P]),
    "sast_postprocess_ws_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "sast_postprocess": (C.c_int, [P, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, P, P, P, P]),
    "sast_yolox_loss_ws_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "sast_yolox_loss": (C.c_int, [P, P, C.POINTER(SastHeadGeom), C.c_int, C.c_int, C.c_int, C.c_int, P, P, P, P, P, P, P]),
    "sast_mask_token_fwd": (C.c_int, [P, P, P, P, C.c_int, C.c_int, C.c_int, P]),
    "sast_mask_token_bwd": (C.c_int, [P, P, P, P, C.c_int, C.c_int, P]),
    "sast_select_pair": (C.c_int, [P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(SastSel), C.POINTER(SastSel), P]),
    "sast_mswsa_bwd_ws_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "sast_mswsa_raw_ws_floats": (C.c_size_t, [C.c_int, C.c_int]),
    "sast_mswsa_fused_ws_floats": (C.c_size_t, [C.c_int] * 5),
    "sast_mswsa_fwd": (C.c_int, [C.POINTER(SastMswsaArgs), P]),
    "sast_mswsa_bwd": (C.c_int, [C.POINTER(SastMswsaArgs), P]),
    "sast_lstm_fwd": (C.c_int, [C.POINTER(SastLstmArgs), P]),
    "sast_lstm_bwd": (C.c_int, [C.POINTER(SastLstmArgs), P]),
    "sast_dwconv_fwd": (C.c_int, [P, P, P, P] + [C.c_int] * 5 + [P]),
    "sast_dwconv_bwd": (C.c_int, [P, P, P, P, P, P] + [C.c_int] * 5 + [P]),
    "sast_conv_bn_ws_floats": (C.c_int, [C.c_int]),
    "sast_conv_bn_silu_fwd": (C.c_int, [C.POINTER(SastConvBnArgs), P]),
    "sast_conv_bn_silu_bwd": (C.c_int, [C.POINTER(SastConvBnArgs), P]),
    "sast_conv_bn_silu2_fwd": (C.c_int, [C.POINTER(SastConvBn2Args), P]),
    "sast_conv_bn_silu2_bwd": (C.c_int, [C.POINTER(SastConvBn2Args), P]),
    "sast_upsample_cat_fwd": (C.c_int, [P, P, P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P]),
    "sast_upsample_cat_bwd": (C.c_int, [P, P, P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P]),
    "sast_cat2_fwd": (C.c_int, [P, P, P, C.c_int, C.c_int, C.c_int, P]),
    "sast_cat2_bwd": (C.c_int, [P, P, P, C.c_int, C.c_int, C.c_int, P]),
    "sast_gather_samples": (C.c_int, [C.POINTER(SastSampleGather), P]),
    "sast_gather_samples_bwd": (C.c_int, [C.POINTER(SastSampleGather), P]),
    "sast_zero_samples": (C.c_int, [P, C.c_int, C.c_size_t, C.POINTER(SastSampleMask), P]),
    "sast_adamw_onecycle": (C.c_int, [P, P, P, P, C.c_size_t, P, C.c_double, C.c_double, F32, F32, F32, F32, C.c_double, C.c_double, C.c_double,
                                      C.c_double, C.c_double, P]),
    "sast_dw_defer": (C.c_int, [C.c_int]),
    "sast_dw_defer_rows": (C.c_int, [C.c_long, C.c_long]),
    "sast_dw_pending": (C.c_int, []),
    "sast_dw_discard": (C.c_int, []),
    "sast_dw_flush": (C.c_int, [P]),
    "sast_launch_count": (C.c_ulonglong, []),
    "sast_config_reload": (C.c_int, []),
    "sast_config_get": (C.c_int, [C.c_char_p, C.c_int]),
    "sast_config_report": (C.c_size_t, [C.c_char_p, C.c_size_t]),
    "sast_prof_enable": (C.c_int, [C.c_int]),
    "sast_prof_calibrate": (C.c_float, [P, C.c_int]),
    "sast_prof_report": (C.c_size_t, [C.c_char_p, C.c_size_t]),
    "sast_adamw": (C.c_int, [P, P, P, P, C.c_size_t, P, C.c_double, C.c_double, F32, F32, F32, F32, P]),
}

_lib = None


def declared_symbols():
    """every function name declared in include/sast_hip.h"""
    with open(HEADER_PATH) as f:
        src = f.read()
    return sorted(set(re.findall(r"\b(sast_[a-z0-9_]+)\s*\(", src)))


def lib():
    global _lib
    if _lib is None:
        # PyTorch-ROCm ships its own libamdhip64.so: it must be in the process BEFORE this library is loaded, otherwise the dynamic
        # linker binds libsast_hip.so to the system copy under /opt/rocm and the process ends up with two HIP runtimes -- torch's
        # owns the device, ours reports "no ROCm-capable device" at the first launch (seen when build() loaded the library first)
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m sast_amd.build` (needs hipcc, gfx950). "
                "sast_amd has no CPU/PyTorch fallback for the SAST hot path.")
        if os.path.abspath(LIB_PATH) != os.path.abspath(DEFAULT_LIB_PATH):
            import sys
            print(f"[sast_amd] WARNING: SAST_LIB_PATH is set: loading {LIB_PATH} instead of the product library {DEFAULT_LIB_PATH}", file=sys.stderr)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def reload_knobs() -> int:
    """the library re-reads every SAST_* tuning knob from the environment at its next use (call after changing os.environ in-process)"""
    return int(lib().sast_config_reload())


def knobs() -> dict:
    """{name: value in use} of the SAST_* knobs the library has read so far"""
    n = lib().sast_config_report(None, 0)
    buf = C.create_string_buffer(n)
    lib().sast_config_report(buf, n)
    out = {}
    for line in buf.value.decode().splitlines():
        k, v = line.split("=", 1)
        out[k] = int(v.split(" ")[0])
    return out


def loaded_path() -> str:
    """the library file behind lib() (the in-tree product unless SAST_LIB_PATH overrides it)"""
    return os.path.abspath(LIB_PATH)


def is_product_library() -> bool:
    return os.path.abspath(LIB_PATH) == os.path.abspath(DEFAULT_LIB_PATH)


_tools = None


def tools_lib():
    """libsast_hip_tools.so: GEMM micro-benchmark / timeline entry points (csrc/k_test.hip) for tools/*.py -- not the product"""
    global _tools
    if _tools is None:
        lib()      # the tools library links against the product library (same directory, rpath $ORIGIN)
        _tools = C.CDLL(os.environ.get("SAST_TOOLS_LIB_PATH") or os.path.join(_HERE, "libsast_hip_tools.so"))   # (variant builds: A/B of k_test.hip)
    return _tools


def check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"libsast_hip: {what} failed with code {rc}")
