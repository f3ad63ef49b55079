#!/bin/bash
# GPU experiment 2: correctness of the rewritten kernel, timeline, a few ablations
cd "$(dirname "$0")/.."
out=gpurun_out/r2c; mkdir -p $out
A="2 128 128 128 32 32 4"; B="2 64 64 64 64 64 4"; C="2 128 128 128 64 32 4"; Dd="2 128 128 128 32 32 2"
PP_VARIANTS=0,1 python scripts/pp_ab.py check > $out/check.log 2>&1; tail -1 $out/check.log
scripts/pp_harness.sh base
scripts/pp_harness.sh trace -DTEM_PP_TRACE
{
echo "== base"; for v in 0 1; do build/pp_harness_base $A $v 10; build/pp_harness_base $B $v 20; build/pp_harness_base $C $v 10; build/pp_harness_base $Dd $v 10 0 0;  build/pp_harness_base $Dd $v 10 0 1; done
echo "== trace A v1"; build/pp_harness_trace $A 1 5
echo "== trace B v1"; build/pp_harness_trace $B 1 5
for abl in 1 2 4 16 31; do
  scripts/pp_harness.sh abl$abl -DTEM_PP_ABL=$abl
  echo "== ablation $abl"; build/pp_harness_abl$abl $A 1 10; build/pp_harness_abl$abl $B 1 20
done
} > $out/exp2.log 2>&1
grep -v "@" $out/exp2.log | tail -30
