set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4d; mkdir -p $O; rm -f $O/*
for v in r4 ; do
  for shp in "2 128 128 128 32 32" "2 128 128 128 64 32" "2 64 64 64 64 64" "2 64 64 64 128 64" "2 32 32 32 128 128"; do
    WG_ONE=3 HARNESS_CHECK_ARITH=1 timeout 120 build/wg_harness_$v $shp 10 3 2>&1 | tail -2 | sed "s/^/$v one=3 /" >> $O/ablate.txt
  done
done
WG_ONE=3 timeout 120 build/wg_harness_r4 2 128 128 128 32 32 10 1 2>&1 | tail -1 | sed "s/^/zs one=3 /" >> $O/ablate.txt
WG_ONE=3 timeout 120 build/wg_harness_r4t 2 128 128 128 32 32 10 3 > $O/trace.txt 2>&1
timeout 1200 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "wgrad" 2>&1 | tail -5 > $O/pytest_wgrad.txt
