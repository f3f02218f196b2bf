#!/bin/bash
# A/B of k_permute_lines build settings on the GPU box (EXTRA flags, one set per argument): config 5 self-overlap.
set -o pipefail
cd "$(dirname "$0")/.."
for extra in "$@"; do
  echo "=== EXTRA=$extra"
  rm -f binary_amd/csrc/query_pipe.o && make -C binary_amd/csrc -s "EXTRA=$extra" 2>&1 | grep -E "error"
  python bench.py --config 5 --no-cpu-baseline --no-extras --no-parity --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  self-overlap ms', d['ms_per_step'])"
done
rm -f binary_amd/csrc/query_pipe.o && make -C binary_amd/csrc -s
