cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4k; mkdir -p $O; rm -f $O/*
for v in r4 r4d r4 r4d; do
  for z in "" 0.6; do
    WG_GZERO=$z WG_ONE=3 timeout 120 build/wg_harness_$v 2 128 128 128 32 32 10 3 2>&1 | tail -1 | sed "s/^/$v gzero=$z /" >> $O/harness.txt
  done
done
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('window', d['ms_per_step'])" >> $O/bench_ab.txt
TEM_LIB=$GRAFT_REPO_ROOT/build/var/libtem_hip_trd.so timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('direct', d['ms_per_step'])" >> $O/bench_ab.txt
done
