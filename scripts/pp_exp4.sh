#!/bin/bash
cd "$(dirname "$0")/.."
out=gpurun_out/r2e; mkdir -p $out
A="2 128 128 128 32 32 4"; B="2 64 64 64 64 64 4"; C="2 128 128 128 64 32 4"; Dd="2 128 128 128 32 32 2"; E="2 32 32 32 128 128 4"
PP_VARIANTS=0,1 python scripts/pp_ab.py check > $out/check.log 2>&1; tail -1 $out/check.log
scripts/pp_harness.sh base
scripts/pp_harness.sh trace -DTEM_PP_TRACE
scripts/pp_harness.sh s0 -DTEM_PP_SCHED=0
{
for tag in base s0; do
  echo "== $tag"; build/pp_harness_$tag $A 1 10; build/pp_harness_$tag $B 1 20; build/pp_harness_$tag $C 1 10; build/pp_harness_$tag $Dd 1 10 0 0; build/pp_harness_$tag $Dd 1 10 0 1; build/pp_harness_$tag $E 1 30
done
echo "== trace A"; build/pp_harness_trace $A 1 5
echo "== trace B"; build/pp_harness_trace $B 1 5
} > $out/exp4.log 2>&1
grep -v "@" $out/exp4.log | tail -16
python -m pytest tests -m gpu -q -x > $out/pytest.log 2>&1; tail -4 $out/pytest.log
python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --kernel-table $out/ktable.txt > $out/bench.json 2> $out/bench.err; cat $out/bench.json | head -c 1500
