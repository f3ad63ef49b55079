cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/x32b; mkdir -p $O
for v in v5_tr v5 v5m13 v4; do
  echo "== $v"
  for shp in "2 128 128 128 32 32" "2 64 64 64 64 64" "2 32 32 32 128 128"; do
    for nr in "1 0" "0 1"; do
      timeout 120 build/zr_harness_$v $shp 1 2 5 $nr 2>&1 | head -12
    done
  done
done > $O/harness_v5.txt 2>&1
