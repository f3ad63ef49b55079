#!/bin/bash
# usage: scripts/pp_harness.sh <tag> "<extra -D flags for conv_pp.hip>"   -> build/pp_harness_<tag>
# (links the harness + a freshly compiled conv_pp.o against the in-tree libtem_hip.so; the executable's conv_pp wins)
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p build
# the harness copy of the kernel gets its own symbol names: a second definition of the same kernel / host stub next to the
# one in libtem_hip.so is ambiguous for the HIP runtime's host-pointer -> kernel map (the library's copy ran)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Wno-unused-result -Dk_conv_pp=k_conv_pp_hx \
    -Dtem_conv_fwd_pp=tem_conv_fwd_pp_hx -Dtem_conv_pp_stat_blocks=tem_conv_pp_stat_blocks_hx -Dtem_pp_trace_buf=tem_pp_trace_buf_hx \
    -Dk_conv_zr=k_conv_zr_hx -Dtem_conv_fwd_zr=tem_conv_fwd_zr_hx -Dtem_conv_zr_stat_blocks=tem_conv_zr_stat_blocks_hx -Dtem_zr_trace_buf=tem_zr_trace_buf_hx \
    $@ scripts/pp_harness.cpp torch_em_amd/csrc/conv_pp.hip torch_em_amd/csrc/conv_zr.hip \
    -Ltorch_em_amd/lib -ltem_hip -Wl,-rpath,$PWD/torch_em_amd/lib -o build/pp_harness_$tag
