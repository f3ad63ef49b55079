cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for rep in 1 2; do for mb in 128 64 32 0; do
echo -n "rep $rep min_mb $mb: "; timeout 300 python bench.py --steps 6 --warmup 3 --precision fp32 --no-cpu-baseline --no-extras --option wgrad_sums_min_mb=$mb 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
done; done
