cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4g; mkdir -p $O; rm -f $O/*
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -30 > $O/pytest_gpu.txt
for z in 1 0 1 0; do
TEM_FUSE_AMAX=$z timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fuse_amax=$z', d['ms_per_step'])" >> $O/bench_ab.txt
done
