set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4c; mkdir -p $O
for one in 3 0 1; do
  for shp in "2 128 128 128 32 32" "2 128 128 128 64 32" "2 64 64 64 64 64" "2 64 64 64 128 64" "2 32 32 32 128 128"; do
    WG_ONE=$one HARNESS_CHECK_ARITH=1 timeout 120 build/wg_harness_r4 $shp 10 3 2>&1 | tail -2 | sed "s/^/one=$one /" >> $O/harness_tr.txt
  done
done
timeout 1200 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "wgrad" 2>&1 | tail -5 > $O/pytest_wgrad.txt
WG_ONE=3 scripts/kernel_power.sh wgrad_tr_one3 timeout 60 build/wg_harness_r4 2 128 128 128 32 32 1200 3 >> $O/power.txt 2>&1
