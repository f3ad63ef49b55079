cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/amp; mkdir -p $O; rm -rf $O/*
(cd /tmp && TEM_PRECISION=amp timeout 600 rocprofv3 --kernel-trace -d $O/rp -o bench --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras > $O/rp.log 2>&1)
f=$(find $O/rp -name "*_kernel_trace.csv" | head -1)
python scripts/step_trace.py $f > $O/step_trace_amp.txt 2>&1
rm -rf $O/rp
grep -c . $O/step_trace_amp.txt
