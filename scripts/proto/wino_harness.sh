#!/bin/bash
# usage: scripts/proto/wino_harness.sh <tag> "<extra -D flags for conv_wino.hip>"   -> build/wino_harness_<tag>
set -e
cd "$(dirname "$0")/../.."
tag=$1; shift
mkdir -p build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $WN_CFLAGS -Wno-unused-result -Dk_conv_wino=k_conv_wino_hx -Dk_pack_wino=k_pack_wino_hx \
    -Dtem_conv_fwd_wino=tem_conv_fwd_wino_hx -Dtem_pack_weights_wino=tem_pack_weights_wino_hx -Dtem_conv_wino_pack_bytes=tem_conv_wino_pack_bytes_hx -Dtem_wn_trace_buf=tem_wn_trace_buf_hx \
    $@ scripts/proto/wino_harness.cpp scripts/proto/conv_wino.hip \
    -Ltorch_em_amd/lib -ltem_hip -Wl,-rpath,$PWD/torch_em_amd/lib -o build/wino_harness_$tag
