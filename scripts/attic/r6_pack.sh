cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/x32a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_unet.py tests/test_gpu_trainer.py -q -k "repack or exact_fp32 or precision" 2>&1 | tail -6
timeout 300 python bench.py --steps 6 --warmup 3 --precision fp32 --no-cpu-baseline --no-extras 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --precision fp32 --no-cpu-baseline --no-extras > /dev/null 2>&1; f=$(find /tmp/pk -name "*kernel_stats.csv" | head -1); head -14 $f | cut -c1-110
