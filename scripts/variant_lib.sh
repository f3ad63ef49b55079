#!/bin/bash
# usage: scripts/variant_lib.sh <tag> <file.hip> "<extra hipcc flags>"  -> build/var/libtem_hip_<tag>.so
# One source of the library recompiled with extra -D flags, linked with the other objects of the in-tree build
# (run `make -C torch_em_amd/csrc` first).  Use with TEM_LIB=build/var/libtem_hip_<tag>.so for A/B timing.
set -e
cd "$(dirname "$0")/.."
tag=$1; src=$2; shift 2
mkdir -p build/var
base=$(basename "$src" .hip)
extra=""
case "$base" in conv_pp|conv_zr|conv_bf16x3|conv_wgrad_tr) extra="-fno-slp-vectorize";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $extra "$@" -c "torch_em_amd/csrc/$base.hip" -o "build/var/${base}_$tag.o"
objs=$(ls build/csrc/*.o | grep -v "/$base.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "build/var/libtem_hip_$tag.so" $objs "build/var/${base}_$tag.o"
echo "build/var/libtem_hip_$tag.so"
