#!/bin/bash
# on the GPU box: the shared-window one-term weight gradient (TEM_TR_SHARE) against the PF2 loop: checks, times, traces
cd $GRAFT_REPO_ROOT; O=gpurun_out/wgsh; mkdir -p $O; rm -f $O/*
for s in "1 16 16 16 32 32" "2 64 64 64 64 64" "1 24 40 40 64 32" "1 8 16 24 32 64"; do
  for one in 1 2; do for st in 0 $one; do
    echo -n "one=$one st=$st $s: " >> $O/check.txt
    HARNESS_CHECK_ARITH=1 WG_ONE=$one WG_ST=$st timeout 120 build/wg_harness_sh $s 3 3 | grep CHECK >> $O/check.txt 2>&1
    echo -n "   (PF2 loop) one=$one st=$st $s: " >> $O/check.txt
    HARNESS_CHECK_ARITH=1 WG_ONE=$one WG_ST=$st timeout 120 build/wg_harness_nosh $s 3 3 | grep CHECK >> $O/check.txt 2>&1
  done; done
done
cat $O/check.txt
for rep in 1 2; do for tag in nosh sh; do for st in 0 1; do
  for shape in "2 128 128 128 32 32" "2 128 128 128 64 32" "2 64 64 64 64 64" "2 32 32 32 128 128"; do
    echo -n "rep$rep $tag st=$st " >> $O/times.txt
    WG_ONE=1 WG_ST=$st WG_GZERO=0.5 timeout 120 build/wg_harness_$tag $shape 20 3 | grep "wgrad\[" >> $O/times.txt
  done
done; done; done
cat $O/times.txt
WG_ONE=1 WG_ST=1 WG_GZERO=0.5 build/wg_harness_noshtrace 2 128 128 128 32 32 10 3 > $O/trace_nosh.txt 2>&1
WG_ONE=1 WG_ST=1 WG_GZERO=0.5 build/wg_harness_shtrace 2 128 128 128 32 32 10 3 > $O/trace_sh.txt 2>&1
head -8 $O/trace_nosh.txt; head -8 $O/trace_sh.txt
