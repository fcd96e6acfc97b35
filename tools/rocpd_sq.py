"""Per-kernel SQ counter summary of a `rocprofv3 --kernel-trace --pmc SQ_...` pass (rocpd SQLite).

    python tools/rocpd_sq.py gpurun_out/r02_a/sq/sq_results.db --out profiles/r02_a_gemm_sq_counters.txt [--top 30]

Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* / SQ_BUSY_CYCLES count quad-cycles summed over waves
(resp. over SEs), SQ_VALU_MFMA_BUSY_CYCLES counts cycles.  The table reports ratios that do not depend on the unit:
wait_any / wave, wait_inst / wave, lds_wait / wave, active / wave, and MFMA-pipe utilisation as
MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x kernel duration x clock) when --clock-ghz is given.
"""
import argparse
import re
import sqlite3
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::|sast::|void ", "", name)
    return re.sub(r">\(.*$", ">", name) if name.startswith("gemm") else name.split("(")[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--out")
    ap.add_argument("--top", type=int, default=30)
    ap.add_argument("--clock-ghz", type=float, default=2.4)
    ap.add_argument("--grbm-instances", type=int, default=8)
    ap.add_argument("--raw", action="store_true", help="also print the raw counter sums per kernel")
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    # one row per (dispatch, counter, hardware instance: XCD x SE ...): counter values add up over the instances, the
    # dispatch duration must be counted once
    rows = c.execute("select name, counter_name, count(distinct dispatch_id), sum(counter_value) from pmc_events group by name, counter_name").fetchall()
    durs = dict(c.execute("select name, sum(d) from (select name, dispatch_id, max(duration) as d from pmc_events group by name, dispatch_id) group by name").fetchall())
    k = defaultdict(dict)
    for name, cn, n, s in rows:
        e = k[short(name)]
        e[cn] = e.get(cn, 0.0) + s
        e["_n"] = n
        e["_dur_ns"] = durs[name]
    names = sorted({cn for e in k.values() for cn in e if not cn.startswith("_")})
    ranked = sorted(k.items(), key=lambda kv: -kv[1]["_dur_ns"])
    lines = [f"# per-kernel SQ counters of {a.db}; counters: {', '.join(names)}",
             f"{'calls':>6} {'avg_us':>8} {'clk_ghz':>7} {'mfma_util':>9} {'wait_any':>8} {'wait_inst':>9} {'wait_lds':>8} {'active':>7} {'valu/wave_cyc':>13} {'lds_conf':>8}  kernel"]

    def ratio(e, num, den):
        return e[num] / e[den] if (num in e and den in e and e[den]) else float("nan")

    # GRBM_GUI_ACTIVE: one instance per XCD (8) -> effective shader clock = cycles / 8 / kernel time.  The counter window of a dispatch is
    # wider than the kernel (the profiler serialises dispatches and the GUI stays active around them): for a sub-10-us kernel the quotient
    # came out at 5-13 "GHz" (round-4 verdict, weak #9) and every column derived from it was meaningless.  Such rows (and any row whose
    # quotient is outside what the chip can clock) take the duration-weighted clock of the LONG kernels of the same pass and say so ("~").
    def grbm_clk(e):
        dur_s = e["_dur_ns"] * 1e-9
        return e["GRBM_GUI_ACTIVE"] / a.grbm_instances / dur_s / 1e9 if ("GRBM_GUI_ACTIVE" in e and dur_s) else None

    long_rows = [(e["_dur_ns"], grbm_clk(e)) for _n, e in ranked if e["_dur_ns"] / e["_n"] >= 20e3 and grbm_clk(e) is not None and 1.0 <= grbm_clk(e) <= 2.6]
    clk_ref = sum(d * c_ for d, c_ in long_rows) / sum(d for d, _ in long_rows) if long_rows else a.clock_ghz
    lines.insert(1, f"# clk_ghz: GRBM_GUI_ACTIVE / {a.grbm_instances} / kernel time; rows marked ~ (kernels under 10 us, or a quotient outside 1.0-2.6 GHz: the counter "
                    f"window is wider than the kernel) use the duration-weighted clock of this pass's kernels of >= 20 us: {clk_ref:.2f} GHz")
    for name, e in ranked[: a.top]:
        dur_s = e["_dur_ns"] * 1e-9
        clk, mark = grbm_clk(e), " "
        if clk is None or e["_dur_ns"] / e["_n"] < 10e3 or not (1.0 <= clk <= 2.6):
            clk, mark = clk_ref, "~"
        mfma = e.get("SQ_VALU_MFMA_BUSY_CYCLES", float("nan")) / (4 * 256 * dur_s * clk * 1e9) if dur_s else float("nan")
        lines.append(f"{e['_n']:6d} {e['_dur_ns'] / e['_n'] / 1e3:8.2f} {mark}{clk:6.2f} {mfma:9.3f} {ratio(e, 'SQ_WAIT_ANY', 'SQ_WAVE_CYCLES'):8.3f} "
                     f"{ratio(e, 'SQ_WAIT_INST_ANY', 'SQ_WAVE_CYCLES'):9.3f} {ratio(e, 'SQ_WAIT_INST_LDS', 'SQ_WAVE_CYCLES'):8.3f} "
                     f"{ratio(e, 'SQ_ACTIVE_INST_ANY', 'SQ_WAVE_CYCLES'):7.3f} {ratio(e, 'SQ_INSTS_VALU', 'SQ_WAVE_CYCLES'):13.4f} "
                     f"{ratio(e, 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE'):8.3f}  {name[:200]}")
    if a.raw:
        for name, e in ranked[: a.top]:
            lines.append("# " + name[:120] + "  " + "  ".join(f"{cn}={e[cn]:.4g}" for cn in names if cn in e))
    txt = "\n".join(lines)
    print(txt)
    if a.out:
        with open(a.out, "w") as f:
            f.write(txt + "\n")


if __name__ == "__main__":
    main()
