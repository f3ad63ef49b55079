#!/bin/bash
# Runs ON THE GPU BOX: exact-fp32 mode (TEM_PRECISION=fp32), working tree against build/base, then the conv tests.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/ab32; mkdir -p $O; rm -rf $O/*
for i in 1 2; do
  (cd build/base && TEM_PRECISION=fp32 timeout 900 python bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base', d['ms_per_step'])") >> $O/ab.txt
  TEM_PRECISION=fp32 timeout 900 python bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new ', d['ms_per_step'])" >> $O/ab.txt
done
cat $O/ab.txt
timeout 1500 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "conv" 2>&1 | tail -3 > $O/pytest_ops.txt
TEM_PRECISION=fp32 timeout 1500 python -m pytest tests/test_gpu_unet.py -q -m gpu -x 2>&1 | tail -3 > $O/pytest_unet_fp32.txt
tail -2 $O/pytest_ops.txt $O/pytest_unet_fp32.txt
