#!/bin/bash
cd "$(dirname "$0")/.."
out=gpurun_out/r2f; mkdir -p $out
A="2 128 128 128 32 32 4"; B="2 64 64 64 64 64 4"; C="2 128 128 128 64 32 4"; Dd="2 128 128 128 32 32 2"; E="2 32 32 32 128 128 4"; F="2 64 64 64 64 128 2"
scripts/pp_harness.sh base
scripts/pp_harness.sh lf0 -DTEM_PP_LF=0
scripts/pp_harness.sh lf1 -DTEM_PP_LF=1
scripts/pp_harness.sh st0 -DTEM_PP_ST_AUX=0
scripts/pp_harness.sh lf0st0 -DTEM_PP_LF=0 -DTEM_PP_ST_AUX=0
scripts/pp_harness.sh rd2 -DTEM_PP_RD=2
scripts/pp_harness.sh prio0 -DTEM_PP_PRIO=0
{
for tag in base lf0 lf1 st0 lf0st0 rd2 prio0 base; do
  echo "== $tag"; build/pp_harness_$tag $A 1 10; build/pp_harness_$tag $B 1 20; build/pp_harness_$tag $C 1 10; build/pp_harness_$tag $Dd 1 10 0 0; build/pp_harness_$tag $E 1 30; build/pp_harness_$tag $F 1 20 0 0
done
} > $out/exp5.log 2>&1
python3 - <<'PY'
import re,collections
rows=collections.OrderedDict(); tag=None
for l in open('gpurun_out/r2f/exp5.log'):
    if l.startswith('=='): tag=l.split()[1]+('' if l.split()[1] not in rows else '2'); rows[tag]=[]; continue
    m=re.search(r"min ([\d.]+) ms",l)
    if m: rows[tag].append(float(m.group(1)))
for t,v in rows.items(): print(f"{t:8s}", ' '.join(f"{x:.4f}" for x in v))
PY
