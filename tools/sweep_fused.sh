#!/bin/bash
# Rebuilds libbivx.so with different single-pass tile shapes and times bench.py for each (run on the GPU box).
set -e
cd "$(dirname "$0")/.."
for cfg in "1024" "512" "256"; do
  set -- $cfg
  make -C binary_amd/csrc -s clean
  make -C binary_amd/csrc -s -j8 EXTRA="-DBIVX_FUSED_THREADS=$1" 2>&1 | grep -E "error" || true
  echo "== threads per tile=$1"
  python tools/quick_modes.py
done
make -C binary_amd/csrc -s clean
make -C binary_amd/csrc -s -j8
