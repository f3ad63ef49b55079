#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the rocprofv3 passes behind profiles/r01_* -- kernel trace + stats, then one --pmc
# pass per TCC counter group and one for the SQ counters (never combined with other tracing domains).
# usage: scripts/collect_profiles.sh <tag>     -> gpurun_out/<tag>/{rocprof,pmc_FETCH_SIZE,pmc_WRITE_SIZE,pmc_sq}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export TEM_BENCH_PREWARM_S=0
CMD="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rocprof -o bench -- $CMD > $OUT/rocprof.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o bench -- $CMD > $OUT/pmc_$c.log 2>&1
done
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq -o bench -- $CMD > $OUT/pmc_sq.log 2>&1
ls $OUT $OUT/*
