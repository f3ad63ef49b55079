#!/bin/bash
# on the GPU box: kernel trace of the SPOCO step of cfg 5 (scripts/bench_workloads.py 5); launches of the LAST step (between the last two k_ema launches) by kernel
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT}
rm -rf /tmp/c5
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/c5 -o c5 -- python $R/scripts/bench_workloads.py 5 > $R/gpurun_out/cfg5_run.log 2>&1
f=$(find /tmp/c5 -name "*_kernel_trace.csv" | head -1)
python - "$f" <<'PY' > $R/gpurun_out/cfg5_last_step.txt
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ema = [i for i, r in enumerate(rows) if "k_ema" in r["Kernel_Name"]]
lo, hi = ema[-2] + 1, ema[-1] + 1
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows[lo:hi]:
    k = r["Kernel_Name"].replace("void ", "").split("(")[0][:70]
    agg[k][0] += 1
    agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"# last step: {hi - lo} launches, {tot / 1e3:.3f} ms of kernel time")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:72s} {n:4d} {t:10.1f} us")
PY
tail -3 $R/gpurun_out/cfg5_run.log; head -50 $R/gpurun_out/cfg5_last_step.txt
