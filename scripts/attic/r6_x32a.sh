cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/x32a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "exact_fp32" 2>&1 | tail -40 > $O/pytest.txt
for o in 1 0; do
timeout 300 python bench.py --steps 6 --warmup 3 --precision fp32 --no-cpu-baseline --no-extras --option fp32_zr=$o > $O/bench_fp32_zr$o.json 2> $O/bench_fp32_zr$o.err
done
cat $O/pytest.txt; grep -o '"ms_per_step": [0-9.]*' $O/bench_fp32_zr*.json
