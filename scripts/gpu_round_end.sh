#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash scripts/gpu_round_end.sh r04'): everything behind profiles/<tag>_*: rocprofv3 kernel
# stats + PMC passes of the bench command, their summaries, the SQ stall counters, the full bench line and the GPU suite.
tag=${1:-r06}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $O
bash scripts/collect_profiles.sh $tag > $O/collect.log 2>&1
for d in rocprof pmc_FETCH_SIZE pmc_WRITE_SIZE pmc_sq; do       # rocprofv3 nests its outputs: bring them to the expected names
  for kind in kernel_stats kernel_trace counter_collection; do
    f=$(find $O/$d -name "*_${kind}.csv" | head -1)
    [ -n "$f" ] && [ "$f" != "$O/$d/bench_${kind}.csv" ] && cp "$f" $O/$d/bench_${kind}.csv
  done
done
python scripts/summarize_profiles.py $O $O/$tag 13 > $O/summarize.log 2>&1
# every launch of the last step in order + the launches under 30 us (default arithmetic, then the fp16-storage mode)
python scripts/step_trace.py $O/rocprof/bench_kernel_trace.csv > $O/${tag}_step_trace.txt 2>&1
for m in amp amp_bf16; do
  rm -rf /tmp/tr_$m
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$m -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --precision $m --no-cpu-baseline --no-extras > $O/trace_$m.log 2>&1)
  f=$(find /tmp/tr_$m -name "*_kernel_trace.csv" | head -1)
  [ -n "$f" ] && python scripts/step_trace.py $f > $O/${tag}_step_trace_$m.txt 2>&1
done
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
# matrix-pipe utilisation and clock of the other arithmetics of the same step (round 6: is a mode at the socket power limit?)
for m in amp fp32; do
  rm -rf /tmp/sq_$m
  C="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --precision $m --no-cpu-baseline --no-extras"
  (cd /tmp && TEM_BENCH_PREWARM_S=0 timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d /tmp/sq_$m -o bench -- $C > $O/sq_$m.log 2>&1)
  python scripts/mfma_busy.py /tmp/sq_$m "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -- $C" > $O/${tag}_mfma_busy_$m.txt 2>&1
done
bash scripts/collect_stalls.sh $tag >> $O/collect.log 2>&1
find $O -name "*.csv" -size +2M -delete
timeout 1500 python bench.py --kernel-table $O/${tag}_bench_kernel_table.txt > $O/${tag}_bench.json 2> $O/bench.err
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -15 > $O/${tag}_pytest_gpu.txt
ls -la $O
