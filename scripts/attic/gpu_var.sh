#!/bin/bash
# Runs ON THE GPU BOX: usage gpu_var.sh <tag> [pytest -k expr]: kernel trace of the bench step with the in-tree library and with
# build/var/libtem_hip_<tag>.so, per-kernel totals side by side, then three alternating bench lines and the tests on the variant.
tag=$1; kexpr=$2
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/var; mkdir -p $O; rm -rf $O/*
V=$GRAFT_REPO_ROOT/build/var/libtem_hip_$tag.so
for w in base $tag; do
  lib=$GRAFT_REPO_ROOT/torch_em_amd/lib/libtem_hip.so; [ $w != base ] && lib=$V
  (cd /tmp && TEM_LIB=$lib timeout 600 rocprofv3 --kernel-trace -d $O/rp_$w -o bench --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras > $O/rp_$w.log 2>&1)
  f=$(find $O/rp_$w -name "*_kernel_trace.csv" | head -1)
  [ -n "$f" ] && python scripts/step_trace.py $f > $O/step_trace_$w.txt 2>&1
done
find $O -name "*.csv" -delete; find $O -name "*.db" -delete
for i in 1 2 3; do
  timeout 600 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base', d['ms_per_step'])" >> $O/ab.txt
  TEM_LIB=$V timeout 600 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'])" >> $O/ab.txt
done
cat $O/ab.txt
[ -n "$kexpr" ] && TEM_LIB=$V timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_unet.py -q -m gpu -x -k "$kexpr" 2>&1 | tail -3
