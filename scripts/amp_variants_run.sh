#!/bin/bash
# on the GPU box: step time of the mixed modes and the default mode with variant builds of conv_wgrad_tr.hip (build/var/libtem_hip_<tag>.so); "base" = in-tree library
cd $GRAFT_REPO_ROOT; O=gpurun_out/ampvar; mkdir -p $O
for rep in 1 2 3; do for tag in base "$@"; do
  if [ $tag = base ]; then unset TEM_LIB; else export TEM_LIB=$PWD/build/var/libtem_hip_$tag.so; fi
  for prec in amp amp_bf16 split16; do
    echo -n "rep$rep $tag $prec " >> $O/times.txt
    TEM_BENCH_PREWARM_S=1 timeout 300 python bench.py --precision $prec --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))" >> $O/times.txt
  done
done; done
cat $O/times.txt
