"""Summarise a rocprofv3 (ROCm 7.2, rocpd SQLite output) kernel trace: per-kernel calls / total / average / share.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [--out profiles/NAME.txt] [--top 40] [--skip-first-frac 0.0]
"""
import argparse
import re
import sqlite3


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = name.replace("sast::", "")
    name = re.sub(r"void ", "", name)
    return name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--out")
    ap.add_argument("--top", type=int, default=45)
    ap.add_argument("--steps", type=int, default=0, help="if given, also print per-step time (total / steps)")
    ap.add_argument("--json", help="with --steps: write {csrc_sha, steps, kernel_ms_per_step, dispatches_per_step} (profiles/kernel_trace_latest.json)")
    args = ap.parse_args()
    c = sqlite3.connect(args.db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name "
                     "order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows)
    lines = [f"# rocprofv3 --kernel-trace --stats summary of {args.db}", f"# total kernel time {tot / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches",
             f"{'calls':>7} {'total_ms':>10} {'avg_us':>9} {'min_us':>8} {'max_us':>8} {'pct':>6}  kernel"]
    for name, n, s, a, mn, mx in rows[: args.top]:
        lines.append(f"{n:7d} {s / 1e6:10.3f} {a / 1e3:9.2f} {mn / 1e3:8.2f} {mx / 1e3:8.2f} {100 * s / tot:6.2f}  {short(name)[:230]}")
    if args.json and not args.steps:      # one input kernel per forward = per step
        args.steps = sum(r[1] for r in rows if "input_prep" in r[0])
    if args.steps:
        lines.append(f"# per step: {tot / 1e6 / args.steps:.3f} ms of kernel time, {sum(r[1] for r in rows) / args.steps:.1f} dispatches")
        if args.json:
            import json, os, sys
            sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            from sast_amd.profiling import csrc_sha
            with open(args.json, "w") as f:
                json.dump({"csrc_sha": csrc_sha(), "steps": args.steps, "kernel_ms_per_step": tot / 1e6 / args.steps,
                           "dispatches_per_step": sum(r[1] for r in rows) / args.steps,
                           "source": "rocprofv3 --kernel-trace of `bench.py --steps 50 --warmup 10` (+ capture and roofline-leg steps): all dispatches / all steps"}, f, indent=1)
    txt = "\n".join(lines)
    print(txt)
    if args.out:
        with open(args.out, "w") as f:
            f.write(txt + "\n")


if __name__ == "__main__":
    main()
