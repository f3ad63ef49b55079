#!/bin/bash
# usage: scripts/wg_harness_fast.sh <tag> "<extra -D flags for conv_wgrad_tr.hip>"   -> build/wg_harness_<tag>
# Same harness as wg_harness.sh for variants of conv_wgrad_tr.hip ONLY: conv_bf16x3.hip and the harness itself are compiled once
# (build/wgh_*.o, 100 s) and every variant costs one 4-second compile + link.
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p build
HIPCC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -fno-slp-vectorize"
REN="-Dk_conv_wgrad_zs=k_conv_wgrad_zs_hx -Dtem_conv_wgrad_bf16x3=tem_conv_wgrad_bf16x3_hx -Dtem_zs_trace_buf=tem_zs_trace_buf_hx -Dtem_tr_trace_buf=tem_tr_trace_buf_hx -Dk_conv_wgrad_tr=k_conv_wgrad_tr_hx -Dtem_conv_wgrad_tr_launch=tem_conv_wgrad_tr_launch_hx"
if [ ! -f build/wgh_bf16x3.o ] || [ torch_em_amd/csrc/conv_bf16x3.hip -nt build/wgh_bf16x3.o ]; then
  $HIPCC $REN -c torch_em_amd/csrc/conv_bf16x3.hip -o build/wgh_bf16x3.o
fi
TRACE=""; case "$*" in *TEM_TR_TRACE*) TRACE="-DTEM_TR_TRACE";; esac
$HIPCC $REN $TRACE -c scripts/wg_harness.cpp -o build/wgh_main_$tag.o
$HIPCC $REN "$@" -c torch_em_amd/csrc/conv_wgrad_tr.hip -o build/wgh_tr_$tag.o
$HIPCC build/wgh_main_$tag.o build/wgh_bf16x3.o build/wgh_tr_$tag.o -Ltorch_em_amd/lib -ltem_hip -Wl,-rpath,'$ORIGIN/../torch_em_amd/lib' -o build/wg_harness_$tag
