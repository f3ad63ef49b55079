#!/bin/bash
# Per-kernel power / clock samples ("settle the power wall with evidence").
# usage: scripts/kernel_power.sh <label> <command ...>   -- runs the command (a harness looping ONE kernel for a few
# seconds) and samples rocm-smi every 100 ms beside it; prints min / median / max of sclk (MHz) and socket power (W)
# over the samples taken while the command was running (first and last 20 % dropped).
label=$1; shift
out=$(mktemp)
( while true; do rocm-smi --showclocks --showpower --csv 2>/dev/null | tr '\n' ' ' ; echo; sleep 0.1; done ) > $out &
spid=$!
"$@"
kill $spid 2>/dev/null; wait $spid 2>/dev/null
python3 - "$label" "$out" <<'PY'
import re, sys, statistics
label, path = sys.argv[1], sys.argv[2]
sclk, pw = [], []
for line in open(path):
    m = re.search(r"\((\d+)Mhz\)", line)          # first clock entry of the row is sclk's current level
    ms = re.findall(r"sclk[^,]*,|", line)
    c = re.findall(r"(\d+)Mhz", line)
    p = re.findall(r"(\d+\.\d+)", line)
    if c: sclk.append([int(v) for v in c])
    if p: pw.append([float(v) for v in p])
print(f"# {label}: {len(sclk)} samples; raw first line: {open(path).readline().strip()[:300]}")
def stats(rows, name):
    if not rows: return
    n = len(rows); rows = rows[n // 5: n - n // 5] or rows
    for j in range(min(len(r) for r in rows)):
        col = [r[j] for r in rows]
        print(f"{label} {name}[{j}]: min {min(col)} median {statistics.median(col)} max {max(col)}")
stats(sclk, "MHz"); stats(pw, "float")
PY
mkdir -p gpurun_out; cp $out gpurun_out/power_raw_$label.txt; rm -f $out
