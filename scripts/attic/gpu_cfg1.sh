cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/cfg1; mkdir -p $O; rm -rf $O/*
cd /tmp; export TMPDIR=/tmp
for c in 1 5; do
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp$c -o b -- python $GRAFT_REPO_ROOT/scripts/bench_workloads.py $c > $O/log$c.txt 2>&1
f=$(find $O/rp$c -name "*kernel_stats.csv" | head -1)
python - "$f" > $O/stats$c.txt <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms over the whole run")
for r in rows[:40]:
    print(f"{r['Name'].replace('void ','').split('(')[0][:90]:90s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.1f} us {float(r['TotalDurationNs'])/1e6:9.3f} ms {float(r['Percentage']):6.2f} %")
PY
rm -rf $O/rp$c
done
