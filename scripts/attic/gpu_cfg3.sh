cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/c3; mkdir -p $O; rm -rf $O/*
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/rp -o b -- python $GRAFT_REPO_ROOT/scripts/bench_workloads.py 3 > $O/log3.txt 2>&1)
f=$(find $O/rp -name "*kernel_trace.csv" | head -1)
python - "$f" > $O/top.txt 2>&1 <<'PY'
import csv,sys,re,collections
rows=list(csv.DictReader(open(sys.argv[1])))
d=collections.defaultdict(lambda:[0,0.0])
for r in rows:
    n=re.sub(r"\(.*","",r["Kernel_Name"]).replace("void ","")[:80]
    d[n][0]+=1; d[n][1]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1000
tot=sum(v[1] for v in d.values())
print("total", tot/1000, "ms,", len(rows), "launches")
for n,(c,t) in sorted(d.items(), key=lambda kv:-kv[1][1])[:40]:
    print(f"{n:82s} {c:5d} {t/c:9.1f} us {t/1000:9.3f} ms")
PY
rm -rf $O/rp; tail -2 $O/log3.txt | head -1; cat $O/top.txt
