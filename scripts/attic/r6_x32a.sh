cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/x32a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_unet.py tests/test_gpu_ops.py -q -s -k "exact_fp32 or wgrad_sums or norm_backward" 2>&1 | tail -30 > $O/pytest.txt
for o in 1 0; do
timeout 300 python bench.py --steps 6 --warmup 3 --precision fp32 --no-cpu-baseline --no-extras --option fp32_zr=$o > $O/bench_fp32_zr$o.json 2> $O/bench_fp32_zr$o.err
done
timeout 300 python bench.py --steps 6 --warmup 3 --precision fp32 --no-cpu-baseline --no-extras --kernel-table $O/ktable_zr1.txt > /dev/null 2>&1
cat $O/pytest.txt; grep -o '"ms_per_step": [0-9.]*' $O/bench_fp32_zr*.json
