cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/dfr; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_unet.py -q -x -k "exact_fp32" 2>&1 | tail -4 > $O/pytest.txt
for v in dfr nodfr dfr44; do
  echo "== $v"
  for shp in "2 128 128 128 32 32" "2 128 128 128 64 32" "2 64 64 64 64 64" "2 32 32 32 128 128"; do
    for nr in "1 0" "0 0"; do
      timeout 120 build/zr_harness_$v $shp 1 2 5 $nr 2>&1 | head -1
    done
  done
done > $O/harness.txt 2>&1
for rep in 1 2; do timeout 300 python bench.py --steps 6 --warmup 3 --precision fp32 --no-cpu-baseline --no-extras 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' >> $O/harness.txt; done
cat $O/pytest.txt $O/harness.txt
