"""usage: step_trace.py <rocprofv3 kernel_trace.csv>  -- the kernels of the LAST optimizer step in launch order with their
durations (end of the previous kernel to end of this one is what a launch costs in a graph replay: the trace shows no gaps),
and the total of the launches shorter than 30 us."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_adamw")]
a, b = idx[-2], idx[-1]
small = {}
tot = 0.0
for r in rows[a + 1:b + 1]:
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:60]
    print(f"{name:62s} {us:8.1f}")
    tot += us
    if us < 30:
        e = small.setdefault(name, [0, 0.0])
        e[0] += 1
        e[1] += us
print(f"\n{b - a} launches, {tot / 1000:.3f} ms")
print("launches under 30 us:")
for name, (n, us) in sorted(small.items(), key=lambda kv: -kv[1][1]):
    print(f"  {name:60s} {n:3d} x {us / n:6.1f} us = {us / 1000:.3f} ms")
print(f"  total {sum(v[0] for v in small.values())} launches, {sum(v[1] for v in small.values()) / 1000:.3f} ms")
