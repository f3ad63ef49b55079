#!/bin/bash
# GPU experiment 3: tap-loop scheduling variants + finer timeline + PMC
cd "$(dirname "$0")/.."
out=gpurun_out/r2d; mkdir -p $out
A="2 128 128 128 32 32 4"; B="2 64 64 64 64 64 4"; C="2 128 128 128 64 32 4"; Dd="2 128 128 128 32 32 2"; E="2 32 32 32 128 128 4"
scripts/pp_harness.sh base
scripts/pp_harness.sh trace -DTEM_PP_TRACE
scripts/pp_harness.sh s1 -DTEM_PP_SCHED=1
scripts/pp_harness.sh s1t -DTEM_PP_SCHED=1 -DTEM_PP_TRACE
scripts/pp_harness.sh rd4 -DTEM_PP_RD=4
scripts/pp_harness.sh s1rd4 -DTEM_PP_SCHED=1 -DTEM_PP_RD=4
scripts/pp_harness.sh s1rd5 -DTEM_PP_SCHED=1 -DTEM_PP_RD=5
{
for tag in base s1 rd4 s1rd4 s1rd5; do
  echo "== $tag"; build/pp_harness_$tag $A 1 10; build/pp_harness_$tag $B 1 20; build/pp_harness_$tag $C 1 10; build/pp_harness_$tag $Dd 1 10 0 0; build/pp_harness_$tag $E 1 30
done
echo "== old kernel"; build/pp_harness_base $A 0 10; build/pp_harness_base $B 0 20; build/pp_harness_base $C 0 10; build/pp_harness_base $Dd 0 10 0 0; build/pp_harness_base $E 0 30
echo "== trace A"; build/pp_harness_trace $A 1 5
echo "== trace B"; build/pp_harness_trace $B 1 5
echo "== trace A s1"; build/pp_harness_s1t $A 1 5
echo "== trace B s1"; build/pp_harness_s1t $B 1 5
} > $out/exp3.log 2>&1
cd /tmp && export TMPDIR=/tmp
for tag in base s1; do
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d /root/repo/$out/pmc_$tag -o pmc --output-format csv -- /root/repo/build/pp_harness_$tag $A 1 5 > /root/repo/$out/pmc_$tag.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d /root/repo/$out/pmcB_$tag -o pmc --output-format csv -- /root/repo/build/pp_harness_$tag $B 1 5 >> /root/repo/$out/pmc_$tag.log 2>&1
done
cd /root/repo; grep -v "@" $out/exp3.log | tail -40; find $out -name "*.csv" | head
