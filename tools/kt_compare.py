"""per-kernel comparison of two rocpd_stats summaries (tools/rocpd_stats.py), normalised per step; LdPlanesOf<X> counts as X.
    python tools/kt_compare.py base.txt steps_base other.txt steps_other [top]"""
import collections
import re
import sys


def load(f, steps):
    d = collections.defaultdict(lambda: [0.0, 0.0])
    for line in open(f):
        m = re.match(r'\s*(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(.*)', line)
        if m:
            name = re.sub(r'LdPlanesOf<(\w+)>', r'\1', m.group(7).strip()).replace(' ', '').split('(')[0]
            d[name][0] += int(m.group(1)) / steps
            d[name][1] += float(m.group(2)) / steps * 1000
    return d


a, b = load(sys.argv[1], float(sys.argv[2])), load(sys.argv[3], float(sys.argv[4]))
top = int(sys.argv[5]) if len(sys.argv) > 5 else 25
print('kernel time per step (us):', round(sum(v[1] for v in a.values())), '->', round(sum(v[1] for v in b.values())))
rows = sorted(((b[n][1] - a[n][1], a[n][0], b[n][0], a[n][1], b[n][1], n[:130]) for n in set(a) | set(b)))
for r in rows[:top]:
    print('%+7.1f us  calls %.1f/%.1f  %7.1f -> %7.1f  %s' % r)
print('...')
for r in rows[-top // 2:]:
    print('%+7.1f us  calls %.1f/%.1f  %7.1f -> %7.1f  %s' % r)
