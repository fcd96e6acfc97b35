"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (rocpd SQLite) per kernel.

    python tools/rocpd_pmc.py --fetch gpurun_out/pmc_f/pf_results.db --write gpurun_out/pmc_w/pw_results.db \
        --out profiles/r01_pmc_hbm_traffic.json

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced
streaming read -> doubled here; WRITE_SIZE is taken as reported (uncalibrated).  Values are KB in the counters.
"""
import argparse
import json
import re
import sqlite3


def short(name):
    name = re.sub(r"\(anonymous namespace\)::|sast::|void ", "", name)
    return name.split("(")[0] if not name.startswith("gemm_kernel") else re.sub(r">\(.*$", ">", name)


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), avg(counter_value), sum(counter_value), avg(duration) from pmc_events where counter_name=? "
                     "group by name", (counter,)).fetchall()
    return {short(n): {"launches": k, "avg_kb": a, "sum_kb": s, "avg_us": d / 1e3} for n, k, a, s, d in rows}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--out")
    ap.add_argument("--top", type=int, default=25)
    a = ap.parse_args()
    f, w = per_kernel(a.fetch, "FETCH_SIZE"), per_kernel(a.write, "WRITE_SIZE")
    out = {}
    for k in f:
        fk, wk = f[k], w.get(k, {"avg_kb": 0.0, "sum_kb": 0.0})
        out[k] = {"launches": fk["launches"], "fetch_kb_raw": fk["avg_kb"], "write_kb_raw": wk["avg_kb"],
                  "hbm_bytes_per_launch": (2.0 * fk["avg_kb"] + wk["avg_kb"]) * 1024.0,
                  "total_bytes": (2.0 * fk["sum_kb"] + wk["sum_kb"]) * 1024.0, "avg_us_profiled": fk["avg_us"]}
    ranked = sorted(out.items(), key=lambda kv: -kv[1]["total_bytes"])
    tot = sum(v["total_bytes"] for v in out.values())
    print(f"total HBM traffic (2*FETCH + WRITE): {tot / 1e9:.3f} GB over the profiled run")
    for k, v in ranked[: a.top]:
        print(f"{v['total_bytes'] / 1e6:10.1f} MB  {v['launches']:6d} launches  {v['hbm_bytes_per_launch'] / 1e6:8.3f} MB/launch  "
              f"{v['hbm_bytes_per_launch'] / (v['avg_us_profiled'] * 1e-6) / 1e12:6.2f} TB/s  {k[:150]}")
    if a.out:
        with open(a.out, "w") as fh:
            import os, sys
            sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            from sast_amd.profiling import csrc_sha
            steps = max([v["launches"] for k, v in out.items() if k.startswith("input_prep_kernel")] or [0])   # one input kernel per forward
            json.dump({"csrc_sha": csrc_sha(), "steps": steps, "correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE under-reports wide reads by 2x)",
                       "total_bytes": tot, "kernels": dict(ranked)}, fh, indent=1)


if __name__ == "__main__":
    main()
