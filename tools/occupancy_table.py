"""What bounds the product kernels (round 6, verdict item 3): per kernel of one step -- static resources (VGPRs + AGPRs, LDS per
workgroup -> waves per SIMD and workgroups per CU; tools/kernel_resources.py), time per launch (rocprofv3 kernel trace), HBM-side
bytes per launch and the rate they imply (PMC passes), that rate against a float4-copy calibration taken in the same gpurun call,
and the matrix-pipe utilisation / wait shares (SQ pass).

    python tools/occupancy_table.py --resources profiles/r06_kernel_resources.json --trace <kernel_trace_stats.txt> --pmc <pmc_hbm_traffic.json>
                                    --sq <sq_counters.txt> --copy-tbs 6.2 --steps N [--top 30] > profiles/r06_z_kernel_occupancy_table.md
"""
import argparse
import json
import re


def key(name):
    """a comparable short form: template arguments kept, the parameter list dropped"""
    name = re.sub(r"\(anonymous namespace\)::|sast::|void ", "", name).strip()
    depth, out = 0, []
    for ch in name:
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return re.sub(r"\s+", "", "".join(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--resources", required=True)
    ap.add_argument("--trace", required=True)
    ap.add_argument("--pmc", required=True)
    ap.add_argument("--sq")
    ap.add_argument("--copy-tbs", type=float, required=True, help="float4 copy rate measured in the same call (read + write bytes / time), TB/s")
    ap.add_argument("--steps", type=int, required=True, help="steps covered by the kernel trace")
    ap.add_argument("--top", type=int, default=30)
    a = ap.parse_args()
    res = {key(k): v for k, v in json.load(open(a.resources))["kernels"].items()}
    pmc = {key(k): v for k, v in json.load(open(a.pmc))["kernels"].items()}
    sq = {}
    if a.sq:
        for line in open(a.sq):
            m = re.match(r"\s*(\d+)\s+([\d.]+)\s+~?\s*([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+\S+\s+(.*)", line)
            if m:
                sq[key(m.group(10))] = {"mfma_util": float(m.group(4)), "wait_any": float(m.group(5)), "wait_inst": float(m.group(6)), "active": float(m.group(8))}
    rows = []
    for line in open(a.trace):
        m = re.match(r"\s*(\d+)\s+([\d.]+)\s+([\d.]+)\s+[\d.]+\s+[\d.]+\s+([\d.]+)\s+(.*)", line)
        if m:
            rows.append((key(m.group(5)), int(m.group(1)), float(m.group(2)), float(m.group(3)), float(m.group(4))))

    def find(table, k):
        if k in table:
            return table[k]
        for kk, v in table.items():          # the trace truncates long names: prefix match
            if kk.startswith(k) or k.startswith(kk):
                return v
        return None

    print(f"float4-copy calibration of this call: {a.copy_tbs:.2f} TB/s (read + write); HBM3E peak 8.0 TB/s (guide: 6.29 TB/s measured for the same copy).")
    print("waves/SIMD = min(8, floor(512 / ceil((VGPR + AGPR) / 8) * 8)); WG/CU = floor(160 KiB / LDS per workgroup); `HBM TB/s` = (2 * FETCH_SIZE + WRITE_SIZE) of the")
    print("PMC passes / the launch's duration in the un-profiled kernel trace; `of copy` = that rate / the calibration; mfma / wait_any / wait_inst / active: shares of the")
    print("wave cycles (SQ pass: matrix pipe busy; parked at s_waitcnt or a barrier; stalled at issue; issuing).\n")
    print("| kernel | launches / step | avg us | % of step | VGPR+AGPR | LDS KB / WG | waves / SIMD | WG / CU (LDS) | HBM MB / launch | HBM TB/s | of copy | mfma | wait_any | wait_inst | active |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for k, calls, total_ms, avg_us, pct in rows[: a.top]:
        r, p, s = find(res, k), find(pmc, k), find(sq, k)
        regs = f"{r['vgpr']}+{r['agpr']}" if r else "?"
        lds = f"{r['lds_bytes_per_workgroup'] / 1024:.1f}" if r else "?"
        wps = r["waves_per_simd_by_registers"] if r else "?"
        wgc = (r["workgroups_per_cu_by_lds"] if r and r["workgroups_per_cu_by_lds"] is not None else "-") if r else "?"
        if isinstance(wgc, int) and wgc > 32:
            wgc = ">32"
        mb = p["hbm_bytes_per_launch"] / 1e6 if p else None
        tbs = (p["hbm_bytes_per_launch"] / (avg_us * 1e-6) / 1e12) if p else None
        f = lambda v, n=2: (f"{v:.{n}f}" if v is not None else "-")
        print(f"| `{k[:150]}` | {calls / a.steps:.1f} | {avg_us:.1f} | {pct:.2f} | {regs} | {lds} | {wps} | {wgc} | {f(mb, 1)} | {f(tbs)} | "
              f"{f(tbs / a.copy_tbs if tbs else None)} | {f(s['mfma_util'] if s else None, 3)} | {f(s['wait_any'] if s else None)} | "
              f"{f(s['wait_inst'] if s else None)} | {f(s['active'] if s else None)} |")


if __name__ == "__main__":
    main()
