#!/bin/bash
# GPU experiment 1: phase timeline + ablations of the ping-pong kernel
cd "$(dirname "$0")/.."
out=gpurun_out/r2b; mkdir -p $out
A="2 128 128 128 32 32 4"; B="2 64 64 64 64 64 4"
scripts/pp_harness.sh base
scripts/pp_harness.sh trace -DTEM_PP_TRACE
{
echo "== base"; for v in 0 1 2; do build/pp_harness_base $A $v 10; done; for v in 0 1; do build/pp_harness_base $B $v 20; done
echo "== trace A v1"; build/pp_harness_trace $A 1 5
echo "== trace A v2"; build/pp_harness_trace $A 2 5
echo "== trace B v1"; build/pp_harness_trace $B 1 5
for abl in 1 2 4 8 16 3 11 15 27 31; do
  scripts/pp_harness.sh abl$abl -DTEM_PP_ABL=$abl
  echo "== ablation $abl"; build/pp_harness_abl$abl $A 1 10; build/pp_harness_abl$abl $A 2 10; build/pp_harness_abl$abl $B 1 20
done
scripts/pp_harness.sh rd4 -DTEM_PP_RD=4; echo "== RD=4"; build/pp_harness_rd4 $A 1 10; build/pp_harness_rd4 $B 1 20
scripts/pp_harness.sh prio0 -DTEM_PP_PRIO=0; echo "== PRIO=0"; build/pp_harness_prio0 $A 1 10; build/pp_harness_prio0 $B 1 20
} > $out/exp1.log 2>&1
tail -3 $out/exp1.log
