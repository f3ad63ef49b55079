set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "wgrad or conv_fwd_dgrad_wgrad or mixed_precision" 2>&1 | tail -15 > $O/pytest_wgrad.txt
for one in 0 1 3; do
  for shp in "2 128 128 128 32 32" "2 128 128 128 64 32" "2 64 64 64 64 64" "2 64 64 64 128 64" "2 32 32 32 128 128"; do
    WG_ONE=$one HARNESS_CHECK_ARITH=1 timeout 120 build/wg_harness_r4 $shp 10 1 2>&1 | tail -2 | sed "s/^/one=$one /" >> $O/harness.txt
  done
done
rocm-smi --showclocks --showpower --csv > $O/smi_idle.txt 2>&1
amd-smi metric --help > $O/amdsmi_help.txt 2>&1
for one in 0 3 1; do
  WG_ONE=$one scripts/kernel_power.sh wgrad_zs_one$one timeout 60 build/wg_harness_r4 2 128 128 128 32 32 1200 1 >> $O/power.txt 2>&1
done
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_f16x2.txt 2>&1
TEM_WGRAD_ARITH=bf16x3 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_bf16x3.txt 2>&1
timeout 1500 python scripts/depth4_error_survey.py > $O/depth4_f16x2.txt 2>&1
