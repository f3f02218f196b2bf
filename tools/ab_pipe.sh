#!/bin/bash
# A/B of query_pipe.hip build settings on the GPU box (EXTRA flags, one set per argument): config 3 both orders.
set -o pipefail
cd "$(dirname "$0")/.."
for extra in "$@"; do
  echo "=== EXTRA=$extra"
  make -C binary_amd/csrc -s clean && make -C binary_amd/csrc -s -j16 "EXTRA=$extra" 2>&1 | grep -E "error"
  BIVX_PIPE=1 timeout -k 10 300 python tools/phase_split.py ${CONFIGS:-3} 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if not l.startswith('{'): continue
    d=json.loads(l)
    for o in ('generated','sorted'):
        r=d[o]; print(' ', d['name'], o, ' '.join(f'{k[:-3]}={r[k]:.4f}' for k in r if k in ('full_ms','count_only_ms')))"
done
make -C binary_amd/csrc -s clean && make -C binary_amd/csrc -s -j16
