#!/bin/bash
# on the GPU box: exact-fp32 step (bench.py --precision fp32) with variant builds of conv_mfma.hip (build/var/libtem_hip_<tag>.so); "base" = in-tree library
cd $GRAFT_REPO_ROOT; O=gpurun_out/fp32var; mkdir -p $O
for rep in 1 2; do for tag in base "$@"; do
  if [ $tag = base ]; then unset TEM_LIB; else export TEM_LIB=$PWD/build/var/libtem_hip_$tag.so; fi
  for opt in "" "--option fwd_persistent=1"; do
    echo -n "rep$rep $tag [$opt] " >> $O/times.txt
    TEM_BENCH_PREWARM_S=1 timeout 300 python bench.py --precision fp32 --steps 5 --warmup 2 --no-extras --no-cpu-baseline $opt 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['roofline']['kernel'], round(d['roofline']['avg_launch_ms'],4))" >> $O/times.txt
  done
done; done
cat $O/times.txt
