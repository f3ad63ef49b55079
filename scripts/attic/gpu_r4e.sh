set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4e; mkdir -p $O; rm -f $O/*
for z in 3 1 3 1; do
TEM_OPT_WGRAD_ZS=$z timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('zs=$z', d['ms_per_step'])" >> $O/bench_ab.txt
done
TEM_OPT_WGRAD_ZS=3 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --precision amp 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('amp zs=3', d['ms_per_step'])" >> $O/bench_ab.txt
TEM_OPT_WGRAD_ZS=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --precision amp 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('amp zs=1', d['ms_per_step'])" >> $O/bench_ab.txt
