#!/bin/bash
# A/B the training-step time on ONE box: scripts/ab_bench.sh VAR "0 1" [reps]  (boxes differ by a few percent)
VAR=$1; VALS=$2; REPS=${3:-3}
for i in $(seq $REPS); do for v in $VALS; do
  env $VAR=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$v', round(d['ms_per_step'],3))"
done; done
