#!/bin/bash
# A/B of k_query_fused build settings on the GPU box: rebuilds libbivx.so with each EXTRA setting given (one per
# argument, quoted), runs the parity tests that exercise the kernel once, then times configs 2/3 (+ 5 sorted with C5=1).
# usage (through gpurun): bash tools/ab_fused.sh "-DBIVX_FUSED_THREADS=512 -DBIVX_FUSED_WAVES=6" "..."
set -o pipefail
cd "$(dirname "$0")/.."
for extra in "$@"; do
  echo "=== EXTRA=$extra"
  make -C binary_amd/csrc -s clean && make -C binary_amd/csrc -s -j8 "EXTRA=$extra" || { echo "build failed"; continue; }
  if [ -z "$NOTEST" ]; then
    timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_errors.py -x -q 2>&1 | tail -2
  fi
  for c in 2 3; do
    python bench.py --config $c --steps 40 --no-cpu-baseline --no-parity 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('config', $c, 'ms', round(b['ms_per_step'],4), 'asc_id', round(b['also_measured']['ascending_id']['ms_per_step'],4), 'sorted_q', round(b['also_measured']['position_sorted_queries']['ms_per_step'],4))"
  done
  if [ -n "$C5" ]; then python tools/config5_sorted.py 2>/dev/null | tail -1; fi
done
make -C binary_amd/csrc -s clean && make -C binary_amd/csrc -s -j8
