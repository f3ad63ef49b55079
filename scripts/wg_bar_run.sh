#!/bin/bash
# on the GPU box: bare s_barrier in the multiplying team (bar) against __syncthreads (sync); barnosh = bar with the PF2 loop
cd $GRAFT_REPO_ROOT; O=gpurun_out/wgbar; mkdir -p $O; rm -f $O/*
for s in "1 16 16 16 32 32" "2 64 64 64 64 64" "1 24 40 40 64 32" "1 8 16 24 32 64" "2 32 32 32 128 128"; do
  for one in 0 1 2 3 4; do
    echo -n "one=$one $s: " >> $O/check.txt
    HARNESS_CHECK_ARITH=1 WG_ONE=$one timeout 120 build/wg_harness_bar $s 3 3 | grep CHECK >> $O/check.txt 2>&1
  done
done
cat $O/check.txt
for rep in 1 2; do for tag in sync bar barnosh; do for cfg in "3 0" "1 0" "1 1" "2 2"; do set -- $cfg
  for shape in "2 128 128 128 32 32" "2 128 128 128 64 32" "2 64 64 64 64 64" "2 32 32 32 128 128"; do
    echo -n "rep$rep $tag one=$1 st=$2 " >> $O/times.txt
    WG_ONE=$1 WG_ST=$2 WG_GZERO=0.5 timeout 120 build/wg_harness_$tag $shape 20 3 | grep "wgrad\[" >> $O/times.txt
  done
done; done; done
cat $O/times.txt
WG_ONE=1 WG_ST=1 WG_GZERO=0.5 build/wg_harness_bartrace 2 128 128 128 32 32 10 3 > $O/trace_bar_one1.txt 2>&1
WG_ONE=3 WG_GZERO=0.5 build/wg_harness_bartrace 2 128 128 128 32 32 10 3 > $O/trace_bar_one3.txt 2>&1
head -8 $O/trace_bar_one1.txt; head -8 $O/trace_bar_one3.txt
