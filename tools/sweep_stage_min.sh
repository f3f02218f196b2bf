#!/bin/bash
# Rebuilds libbivx.so with different thresholds for the LDS-staged output and times configs 2 and 3 (GPU box).
set -e
cd "$(dirname "$0")/.."
for v in ${STAGE_MINS:-64 160 320}; do
  make -C binary_amd/csrc -s clean
  make -C binary_amd/csrc -s -j8 EXTRA="-DBIVX_STAGE_MIN=$v" 2>&1 | grep -E "error" || true
  echo "== stage min $v"
  python tools/quick_modes.py 2>&1 | grep ordered
  python tools/measure_configs.py 2>/dev/null | python -c "
import json,sys
s=sys.stdin.read(); d=json.loads(s[s.index('{'):])
for c in d['results']:
    print('  ', c['name'], {o:(round(c[o]['single_pass_ms'],4), round(c[o]['unordered_begin_count_ms'],4)) for o in ('random','sorted')})"
done
make -C binary_amd/csrc -s clean
make -C binary_amd/csrc -s -j8
