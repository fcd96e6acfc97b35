"""Static resources of every kernel of the product library (round 6, verdict item 3): VGPRs / AGPRs / SGPRs / LDS bytes / scratch and
the waves per SIMD they allow, from hipcc's own `-Rpass-analysis=kernel-resource-usage` remarks (no GPU needed).

    python tools/kernel_resources.py [--out profiles/r06_kernel_resources.json]

waves per SIMD: min(8, floor(512 / alloc)) with alloc = ceil((VGPRs + AGPRs) / 8) * 8 (unified register file, granule 8;
MI355X_MICROARCH.md "Register files"); workgroups per CU by LDS: floor(160 KiB / LDS per workgroup)."""
import argparse
import concurrent.futures as cf
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sast_amd import build as B  # noqa: E402


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return p.stdout.split("\n")[: len(names)]


def one(src):
    cmd = [B._hipcc(), *B.FLAGS, "-c", os.path.join(B.CSRC, src), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    out, cur = {}, None
    for line in err.split("\n"):
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {"source": src})
            continue
        m = re.search(r"remark:\s+([A-Za-z /\[\]]+): (\S+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_kernel_resources.json"))
    a = ap.parse_args()
    with cf.ThreadPoolExecutor(max_workers=4) as ex:
        parts = list(ex.map(one, B.SOURCES))
    raw = {}
    for p in parts:
        raw.update(p)
    names = list(raw)
    res = {}
    for mangled, nice in zip(names, demangle(names)):
        r = raw[mangled]
        v, ag = int(r.get("VGPRs", 0)), int(r.get("AGPRs", 0))
        alloc = (v + ag + 7) // 8 * 8
        lds = int(r.get("LDS Size [bytes/block]", 0))
        nice = re.sub(r"\(anonymous namespace\)::|sast::|void ", "", nice)
        res[nice] = {"source": r["source"], "vgpr": v, "agpr": ag, "sgpr": int(r.get("TotalSGPRs", 0)), "scratch_bytes_per_lane": int(r.get("ScratchSize [bytes/lane]", 0)),
                     "lds_bytes_per_workgroup": lds, "waves_per_simd_by_registers": min(8, 512 // max(alloc, 8)),
                     "waves_per_simd_compiler": int(r.get("Occupancy [waves/SIMD]", 0)), "workgroups_per_cu_by_lds": (160 * 1024 // lds) if lds else None}
    from sast_amd.profiling import csrc_sha
    with open(a.out, "w") as f:
        json.dump({"csrc_sha": csrc_sha(), "flags": B.FLAGS, "kernels": res}, f, indent=1, sort_keys=True)
    print(len(res), "kernels ->", a.out)


if __name__ == "__main__":
    main()
