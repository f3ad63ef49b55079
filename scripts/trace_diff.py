"""usage: trace_diff.py <step_trace_a.txt> <step_trace_b.txt>: per-kernel totals of two scripts/step_trace.py outputs side by side"""
import collections
import re
import sys


def load(f):
    d = collections.OrderedDict()
    for ln in open(f):
        m = re.match(r"^(\S.*?)\s+([\d.]+)$", ln.rstrip())
        if not m or ln.startswith(" "):
            continue
        e = d.setdefault(m.group(1).strip(), [0, 0.0])
        e[0] += 1
        e[1] += float(m.group(2))
    return d


a, b = load(sys.argv[1]), load(sys.argv[2])
rows = []
for k in dict.fromkeys(list(a) + list(b)):
    x, y = a.get(k, [0, 0.0]), b.get(k, [0, 0.0])
    rows.append((y[1] - x[1], k, x, y))
for d, k, x, y in sorted(rows):
    if abs(d) > 3:
        print(f"{k[:58]:58s} a {x[0]:3d} x {x[1]:8.1f}   b {y[0]:3d} x {y[1]:8.1f}   diff {d:+8.1f}")
