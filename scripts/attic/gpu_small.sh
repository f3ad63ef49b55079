#!/bin/bash
# Runs ON THE GPU BOX: the small-launch work of round 4 -- the tests that cover it, a bench line, and a kernel trace of
# the bench command (per-kernel durations of the last step: scripts/step_trace.py).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/small; mkdir -p $O; rm -rf $O/*
timeout 1200 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "norm or sums or absmax or amax or wgrad_fp16 or stats" 2>&1 | tail -5 > $O/pytest_ops.txt
timeout 1500 python -m pytest tests/test_gpu_unet.py tests/test_gpu_trainer.py -q -m gpu -x 2>&1 | tail -5 > $O/pytest_unet.txt
timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/bench.json 2> $O/bench.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/rocprof -o bench --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/rocprof.log 2>&1)
f=$(find $O/rocprof -name "*_kernel_trace.csv" | head -1)
[ -n "$f" ] && python scripts/step_trace.py $f > $O/step_trace.txt 2>&1
find $O -name "*.csv" -size +3M -delete; find $O -name "*.db" -delete
tail -3 $O/pytest_ops.txt $O/pytest_unet.txt; cat $O/bench.json | head -c 600
