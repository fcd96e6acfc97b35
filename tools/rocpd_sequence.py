"""One replayed step of a rocprofv3 kernel trace in execution order: start offset, duration, gap to the previous kernel's end, name.

    python tools/rocpd_sequence.py /tmp/kt/kt_results.db --launches 280 [--out profiles/NAME.txt]
Takes the LAST complete window of `launches` dispatches of the trace (the graph replays at the end of a bench run)."""
import argparse, re, sqlite3


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name).replace("sast::", "")
    return re.sub(r"void ", "", name).split("(")[0]


ap = argparse.ArgumentParser()
ap.add_argument("db"); ap.add_argument("--launches", type=int, default=280); ap.add_argument("--out")
a = ap.parse_args()
c = sqlite3.connect(a.db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
st, en = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
rows = c.execute(f"select {st}, {en}, name from kernels order by {st}").fetchall()
rows = rows[-a.launches:]
t0 = rows[0][0]
lines = [f"# last {len(rows)} dispatches of {a.db}: start_us  dur_us  gap_us  kernel"]
prev = None; gaps = 0.0; durs = 0.0
for s, e, n in rows:
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    gaps += max(gap, 0.0); durs += (e - s) / 1e3
    lines.append(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.2f} {gap:7.2f}  {short(n)[:150]}")
    prev = e
lines.insert(1, f"# span {(rows[-1][1] - t0) / 1e3:.1f} us, sum of durations {durs:.1f} us, sum of positive gaps {gaps:.1f} us")
txt = "\n".join(lines)
print(txt)
if a.out:
    open(a.out, "w").write(txt + "\n")
