cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT}
rm -rf /tmp/c1
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/c1 -o c1 -- python $R/scripts/bench_workloads.py 1 > $R/gpurun_out/cfg1_run.log 2>&1
f=$(find /tmp/c1 -name "*_kernel_trace.csv" | head -1)
python - "$f" <<'PY' > $R/gpurun_out/cfg1_last_step.txt
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ad = [i for i, r in enumerate(rows) if "k_adamw" in r["Kernel_Name"]]
lo, hi = ad[-2] + 1, ad[-1] + 1
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows[lo:hi]:
    k = r["Kernel_Name"].replace("void ", "").split("(")[0][:90]
    agg[k][0] += 1
    agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
span = (int(rows[hi-1]["End_Timestamp"]) - int(rows[lo]["Start_Timestamp"])) / 1e6
print(f"# last step: {hi - lo} launches, {tot / 1e3:.3f} ms of kernel time, span {span:.3f} ms")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:92s} {n:4d} {t:10.1f} us")
PY
tail -2 $R/gpurun_out/cfg1_run.log; head -45 $R/gpurun_out/cfg1_last_step.txt
