#!/bin/bash
# Runs ON THE GPU BOX: same-box A/B of the working tree against build/base (a `git archive HEAD` copy with its own library):
# bench lines alternating, then a kernel trace of each (scripts/step_trace.py), then the tests that cover the change.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/ab; mkdir -p $O; rm -rf $O/*
for i in 1 2 3; do
  (cd build/base && timeout 600 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base', d['ms_per_step'])") >> $O/ab.txt
  timeout 600 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new ', d['ms_per_step'])" >> $O/ab.txt
done
cat $O/ab.txt
for w in base new; do
  d=$GRAFT_REPO_ROOT; [ $w = base ] && d=$GRAFT_REPO_ROOT/build/base
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $O/rocprof_$w -o bench --output-format csv -- python $d/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras > $O/rocprof_$w.log 2>&1)
  f=$(find $O/rocprof_$w -name "*_kernel_trace.csv" | head -1)
  [ -n "$f" ] && python scripts/step_trace.py $f > $O/step_trace_$w.txt 2>&1
done
find $O -name "*.csv" -delete; find $O -name "*.db" -delete
timeout 1500 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "norm or sums or absmax or amax or wgrad_fp16 or stat or split_k or maxpool" 2>&1 | tail -3 > $O/pytest_ops.txt
timeout 1500 python -m pytest tests/test_gpu_unet.py tests/test_gpu_spoco.py tests/test_gpu_trainer.py -q -m gpu -x 2>&1 | tail -3 > $O/pytest_unet.txt
tail -2 $O/pytest_ops.txt $O/pytest_unet.txt
