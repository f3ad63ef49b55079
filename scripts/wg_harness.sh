#!/bin/bash
# usage: scripts/wg_harness.sh <tag> "<extra -D flags for conv_bf16x3.hip>"   -> build/wg_harness_<tag>
# the trace (-DTEM_ZS_TRACE), ablation (-DTEM_ZS_ABL=n), -DTEM_ZS_PIPE / -DTEM_ZS_SWAP builds need scripts/wg_experiments.patch applied
# (git apply scripts/wg_experiments.patch): the measured-and-rejected staging/MFMA overlap variants of the wgrad kernel
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -Dk_conv_wgrad_zs=k_conv_wgrad_zs_hx \
    -Dtem_conv_wgrad_bf16x3=tem_conv_wgrad_bf16x3_hx -Dtem_zs_trace_buf=tem_zs_trace_buf_hx -Dtem_tr_trace_buf=tem_tr_trace_buf_hx $@ -Dk_conv_wgrad_tr=k_conv_wgrad_tr_hx -Dtem_conv_wgrad_tr_launch=tem_conv_wgrad_tr_launch_hx scripts/wg_harness.cpp torch_em_amd/csrc/conv_bf16x3.hip torch_em_amd/csrc/conv_wgrad_tr.hip \
    -Ltorch_em_amd/lib -ltem_hip -Wl,-rpath,$PWD/torch_em_amd/lib -o build/wg_harness_$tag
