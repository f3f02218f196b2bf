#!/bin/bash
# A/B of k_query_pipe's workgroup size on the GPU box ("flags|wgs" per argument): configs 3 and 2 through bench.py.
set -o pipefail
cd "$(dirname "$0")/.."
for arg in "$@"; do
  extra="${arg%%|*}"; wgs="${arg##*|}"
  echo "=== EXTRA=$extra WGS=$wgs"
  rm -f binary_amd/csrc/query_pipe.o && make -C binary_amd/csrc -s "EXTRA=$extra" 2>&1 | grep -E "error"
  if [ -n "$wgs" ]; then export BIVX_PIPE_WGS=$wgs; else unset BIVX_PIPE_WGS; fi
  for c in 3 2; do
    python bench.py --config $c --no-cpu-baseline --no-extras --steps 60 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  config', $c, d['ms_per_step'], d['parity'])"
  done
  python bench.py --config 3 --sorted-queries --no-cpu-baseline --no-extras --no-parity --steps 60 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  config 3 sorted', d['ms_per_step'])"
done
unset BIVX_PIPE_WGS
rm -f binary_amd/csrc/query_pipe.o && make -C binary_amd/csrc -s
