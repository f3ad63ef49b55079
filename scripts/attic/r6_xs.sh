cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/xs; mkdir -p $O
for z in 2 1; do
  echo "== fp32_zr=$z"
  for shp in "2 128 128 128 32 32" "2 128 128 128 64 32" "2 64 64 64 64 64" "2 64 64 64 128 64" "2 32 32 32 128 128" "2 32 32 32 256 128"; do
    for nr in "1 0" "0 1"; do
      FP32_ZR=$z timeout 120 build/zr_harness_xs $shp 1 2 5 $nr 2>&1 | head -3
    done
  done
done > $O/harness_xs.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "exact_fp32" 2>&1 | tail -5 >> $O/harness_xs.txt
