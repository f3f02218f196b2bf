#!/bin/bash
# A/B of k_query_pipe_ms build settings on the GPU box (EXTRA flags | BIVX_PIPE_WGS, one pair per argument "flags|wgs").
set -o pipefail
cd "$(dirname "$0")/.."
for arg in "$@"; do
  extra="${arg%%|*}"; wgs="${arg##*|}"
  echo "=== EXTRA=$extra WGS=$wgs"
  rm -f binary_amd/csrc/query_pipe.o && make -C binary_amd/csrc -s "EXTRA=$extra" 2>&1 | grep -E "error"
  for m in 1e4 1e5; do
    if [ -n "$wgs" ]; then export BIVX_PIPE_WGS=$wgs; else unset BIVX_PIPE_WGS; fi
    python tools/skewed_bench.py $m 2>&1 | grep maxlen | cut -c1-40,95-200
  done
done
unset BIVX_PIPE_WGS
rm -f binary_amd/csrc/query_pipe.o && make -C binary_amd/csrc -s
