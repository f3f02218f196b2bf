#!/bin/bash
# Builds binary_amd/libbivx.so.<tag> with extra compiler flags for the listed translation units (the others come from the
# normal build), for A/B runs on the GPU box without compiling there: BIVX_LIB=binary_amd/libbivx.so.<tag> python ...
# usage: tools/build_variant.sh <tag> "<extra flags>" [unit ...]      (units default to query_pipe)
set -eo pipefail
TAG=$1; EXTRA=$2; shift; shift
UNITS=${*:-query_pipe}
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/binary_amd/csrc
make -C "$C" -s -j4 all
D=/tmp/bivx_variants/$TAG
mkdir -p "$D"
OBJS=""
for o in scan build query query_fused query_pipe capi sharded; do
  if [[ " $UNITS " == *" $o "* ]]; then
    src=$C/$o.hip; xl=""
    [[ -f $src ]] || { src=$C/$o.cpp; xl="-x hip"; }
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wextra -Wno-unused-parameter $EXTRA $xl -c "$src" -o "$D/$o.o"
    OBJS="$OBJS $D/$o.o"
  else
    OBJS="$OBJS $C/$o.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/binary_amd/libbivx.so.$TAG" $OBJS $(grep -o '\-lrccl' "$C/Makefile" | head -1)
echo "built binary_amd/libbivx.so.$TAG ($EXTRA; $UNITS)"
