cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4j; mkdir -p $O; rm -f $O/*
timeout 1500 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "conv_fwd_dgrad_wgrad or wgrad" 2>&1 | tail -5 > $O/pytest.txt
for shp in "2 128 128 128 32 32" "2 64 64 64 64 64" "2 32 32 32 128 128"; do
  WG_ONE=4 HARNESS_CHECK_ARITH=1 timeout 120 build/wg_harness_r4 $shp 5 3 2>&1 | tail -2 >> $O/harness_fp32.txt
done
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --precision fp32 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32 tr', d['ms_per_step'])" >> $O/bench_fp32.txt
TEM_OPT_WGRAD_ZS=1 timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --precision fp32 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32 old', d['ms_per_step'])" >> $O/bench_fp32.txt
timeout 900 python -m pytest tests/test_gpu_unet.py -q -m gpu -x -k "golden or mfma_sizes" 2>&1 | tail -3 >> $O/pytest.txt
TEM_PRECISION=fp32 timeout 900 python -m pytest tests/test_gpu_unet.py -q -m gpu -x -k "golden or mfma_sizes" 2>&1 | tail -3 >> $O/pytest.txt
