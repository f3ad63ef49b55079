#!/bin/bash
# Build an A/B variant of libtem_hip.so with extra -D flags for ONE source file, into build/ablibs/<name>.so
# usage: scripts/ab_lib.sh <name> <file.hip> "<-DFLAG=... ...>"   then run with TEM_LIB=build/ablibs/<name>.so
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; FILE=$2; FLAGS=$3
mkdir -p $ROOT/build/ablibs/$NAME
cd $ROOT/torch_em_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $FLAGS -c $FILE -o $ROOT/build/ablibs/$NAME/${FILE%.hip}.o
OBJS=""
for f in capi conv conv_mfma conv_bf16x3 conv_small wgrad_sums norm pool_upsample dice optim label spoco augment predict; do
  if [ "$f.hip" == "$FILE" ]; then OBJS="$OBJS $ROOT/build/ablibs/$NAME/$f.o"; else OBJS="$OBJS $ROOT/build/csrc/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/build/ablibs/$NAME.so $OBJS
echo built $ROOT/build/ablibs/$NAME.so
