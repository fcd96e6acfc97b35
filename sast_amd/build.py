"""Build libsast_hip.so (gfx950) in-tree with hipcc.  `python -m sast_amd.build [--force]`.

Two libraries: libsast_hip.so (the product: every C-ABI entry point of include/sast_hip.h) and libsast_hip_tools.so (the
GEMM micro-benchmark / timeline entry points of csrc/k_test.hip used by tools/*.py only; links against the product library)."""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libsast_hip.so")
TOOLS_LIB = os.path.join(HERE, "libsast_hip_tools.so")
BF16_LIB = os.path.join(HERE, "libsast_hip_bf16.so")
ARCH = "gfx950"
SOURCES = ["k_rows.hip", "k_select.hip", "k_attn_mfma.hip", "k_mswsa_fused.hip", "k_block.hip", "k_conv.hip", "k_dwconv.hip", "k_prof.hip", "k_head.hip", "k_defer.hip", "k_config.hip"]
TOOLS_SOURCES = ["k_test.hip", "k_dma_test.hip", "k_ws_test.hip"]
HEADERS = ["common.cuh", "gemm.cuh", "mfma_tiles.cuh", "gemm_dispatch.cuh", "kernels.h", os.path.join("..", "..", "include", "sast_hip.h")]
FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"] + os.environ.get("SAST_EXTRA_FLAGS", "").split()


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the SAST MI355X kernels cannot be built")


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def _compile(src: str, force: bool, flags=None, obj_dir=None) -> str:
    obj_dir = obj_dir or OBJ
    obj = os.path.join(obj_dir, os.path.splitext(src)[0] + ".o")
    deps = [os.path.join(CSRC, src)] + [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= _newest(deps):
        return obj
    cmd = [_hipcc(), *(FLAGS + list(flags or [])), "-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-6000:]}")
    return obj


def _link(objs, out, force, extra=()):
    if force or not os.path.exists(out) or os.path.getmtime(out) < _newest(objs):
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", out, *extra]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    srcs = SOURCES + TOOLS_SOURCES
    with cf.ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    _link(objs[:len(SOURCES)], LIB, force)
    _link(objs[len(SOURCES):], TOOLS_LIB, force or os.path.getmtime(TOOLS_LIB) < os.path.getmtime(LIB) if os.path.exists(TOOLS_LIB) else True,
          extra=["-L" + HERE, "-lsast_hip", "-Wl,-rpath,$ORIGIN"])
    # the reduced-precision variant of the product library (bf16 GEMM operands, `bench.py --precision bf16`): same sources, one flag.
    # OPT-IN (SAST_BUILD_BF16=1 or `python -m sast_amd.build --bf16`): not part of the product, a compile failure there must not fail build()
    if os.environ.get("SAST_BUILD_BF16", "0") == "1":
        stale = force or not os.path.exists(BF16_LIB) or os.path.getmtime(BF16_LIB) < os.path.getmtime(LIB)
        if stale:
            build_variant(BF16_LIB, ["-DSAST_MFMA_BF16=1"], obj_dir=os.path.join(OBJ, "bf16"))
    if verbose:
        print("built", LIB, "and", TOOLS_LIB)
    return LIB


def build_variant(out: str, flags, obj_dir=None) -> str:
    """an A/B build of the product library with extra compile flags (e.g. -DSAST_SINGLE_TILE_ACCS=1) next to the in-tree one;
    run with SAST_LIB_PATH=<out> (tools only: A/B measurements inside one gpurun call)."""
    out = os.path.abspath(out)
    obj_dir = obj_dir or out + ".obj"
    os.makedirs(obj_dir, exist_ok=True)
    with cf.ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(lambda s: _compile(s, True, flags, obj_dir), SOURCES))
    _link(objs, out, True)
    return out


if __name__ == "__main__":
    if "--out" in sys.argv:
        o = sys.argv[sys.argv.index("--out") + 1]
        fl = sys.argv[sys.argv.index("--flags") + 1].split() if "--flags" in sys.argv else []
        print("built", build_variant(o, fl))
    else:
        if "--bf16" in sys.argv:
            os.environ["SAST_BUILD_BF16"] = "1"
        build(force="--force" in sys.argv, verbose=True)
