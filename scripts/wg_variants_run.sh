#!/bin/bash
# on the GPU box: every build/wg_harness_<tag> given, fp16 2x1 arithmetic (WG_ONE=3), three layer shapes, alternating order x 2
cd $GRAFT_REPO_ROOT; O=gpurun_out/wgv; mkdir -p $O
for rep in 1 2; do for tag in "$@"; do
  for shape in "2 128 128 128 32 32" "2 128 128 128 64 32" "2 64 64 64 64 64" "2 32 32 32 128 128"; do
    echo -n "rep$rep $tag " >> $O/times.txt
    WG_ONE=3 WG_GZERO=0.5 timeout 120 build/wg_harness_$tag $shape 20 3 | grep "wgrad\[" >> $O/times.txt
  done
done; done
cat $O/times.txt
