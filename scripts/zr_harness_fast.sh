#!/bin/bash
# usage: scripts/zr_harness_fast.sh <tag> <extra -D flags for conv_zr.hip>   -> build/zr_harness_<tag>
# pp_harness.cpp + conv_pp.hip are compiled once (build/zrh_*.o); a variant costs one compile of conv_zr.hip (~40 s)
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p build
HIPCC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -fno-slp-vectorize"
REN="-Dk_conv_pp=k_conv_pp_hx -Dtem_conv_fwd_pp=tem_conv_fwd_pp_hx -Dtem_conv_pp_stat_blocks=tem_conv_pp_stat_blocks_hx -Dtem_pp_trace_buf=tem_pp_trace_buf_hx -Dk_conv_zr=k_conv_zr_hx -Dtem_conv_fwd_zr=tem_conv_fwd_zr_hx -Dtem_conv_zr_stat_blocks=tem_conv_zr_stat_blocks_hx -Dtem_zr_trace_buf=tem_zr_trace_buf_hx"
[ -f build/zrh_pp.o ] && [ build/zrh_pp.o -nt torch_em_amd/csrc/conv_pp.hip ] || $HIPCC $REN -c torch_em_amd/csrc/conv_pp.hip -o build/zrh_pp.o
[ -f build/zrh_main.o ] && [ build/zrh_main.o -nt scripts/pp_harness.cpp ] || $HIPCC $REN -c scripts/pp_harness.cpp -o build/zrh_main.o
$HIPCC $REN "$@" -c torch_em_amd/csrc/conv_zr.hip -o build/zrh_zr_$tag.o
$HIPCC build/zrh_main.o build/zrh_pp.o build/zrh_zr_$tag.o -Ltorch_em_amd/lib -ltem_hip -Wl,-rpath,'$ORIGIN/../torch_em_amd/lib' -o build/zr_harness_$tag
