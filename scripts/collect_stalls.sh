#!/bin/bash
# Runs ON THE GPU BOX: the SQ issue / wait / LDS counters behind profiles/r0N_pmc_stalls.txt, one rocprofv3 --pmc pass per
# counter group (never combined with other tracing domains).  usage: scripts/collect_stalls.sh <tag> -> gpurun_out/<tag>/stalls/*
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1/stalls
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export TEM_BENCH_PREWARM_S=0
CMD="python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras ${TEM_STALLS_ARGS:-}"   # TEM_STALLS_ARGS="--precision amp": the 16-bit-storage step
i=0
for grp in "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout 900 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -o b -- $CMD > $OUT/g$i.log 2>&1
  f=$(find $OUT/g$i -name "b_counter_collection.csv" | head -1)
  [ -n "$f" ] && [ "$f" != "$OUT/g$i/b_counter_collection.csv" ] && cp $f $OUT/g$i/b_counter_collection.csv
done
python $R/scripts/pmc_stalls.py $OUT > $R/gpurun_out/$1/pmc_stalls.txt 2>&1
rm -rf $OUT/g*/*/   # keep only the copied csv + logs (size)
