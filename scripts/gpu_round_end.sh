#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash scripts/gpu_round_end.sh r04'): everything behind profiles/<tag>_*: rocprofv3 kernel
# stats + PMC passes of the bench command, their summaries, the SQ stall counters, the full bench line and the GPU suite.
tag=${1:-r04}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$tag; mkdir -p $O
bash scripts/collect_profiles.sh $tag > $O/collect.log 2>&1
for d in rocprof pmc_FETCH_SIZE pmc_WRITE_SIZE pmc_sq; do       # rocprofv3 nests its outputs: bring them to the expected names
  for kind in kernel_stats kernel_trace counter_collection; do
    f=$(find $O/$d -name "*_${kind}.csv" | head -1)
    [ -n "$f" ] && [ "$f" != "$O/$d/bench_${kind}.csv" ] && cp "$f" $O/$d/bench_${kind}.csv
  done
done
python scripts/summarize_profiles.py $O $O/$tag 13 > $O/summarize.log 2>&1
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
bash scripts/collect_stalls.sh $tag >> $O/collect.log 2>&1
find $O -name "*.csv" -size +2M -delete
timeout 1500 python bench.py --kernel-table $O/${tag}_bench_kernel_table.txt > $O/${tag}_bench.json 2> $O/bench.err
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -15 > $O/${tag}_pytest_gpu.txt
ls -la $O
