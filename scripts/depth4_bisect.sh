#!/bin/bash
# depth-4 gradient error against float64 (scripts/depth4_error_survey.py) on
# seeds 0 and 5 with each fusion of the engine switched off in turn, for the exact-fp32 build and the default arithmetic.
# usage (on the GPU box): scripts/depth4_bisect.sh > profiles/r04_depth4_bisect.txt
cd "$(dirname "$0")/.."
for prec in fp32 split16; do
  for sw in "" "TEM_FUSE_STATS=0" "TEM_FUSE_CONCAT_STATS=0" "TEM_OPT_WGRAD_SUMS=0" "TEM_DEFER_CONCAT_NORM=0" "TEM_FUSE_NORM_BWD_DGRAD=0" \
            "TEM_FUSE_STATS=0 TEM_FUSE_CONCAT_STATS=0 TEM_OPT_WGRAD_SUMS=0 TEM_DEFER_CONCAT_NORM=0 TEM_FUSE_NORM_BWD_DGRAD=0" "TEM_WGRAD_ARITH=bf16x3"; do
    echo "## TEM_PRECISION=$prec  ${sw:-(all fusions on: product default)}"
    env TEM_PRECISION=$prec $sw python scripts/depth4_error_survey.py 0 5 2>&1 | grep -v "^#"
  done
done
