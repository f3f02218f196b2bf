#!/bin/bash
# A/B of build.hip compile-time settings on the GPU box (EXTRA flags, one set per argument): tools/build_bench.py + the
# kernel times of one traced run.
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for extra in "$@"; do
  echo "=== EXTRA=$extra"
  rm -f binary_amd/csrc/build.o && make -C binary_amd/csrc -s "EXTRA=$extra" 2>&1 | grep -E "error"
  timeout -k 10 300 python tools/build_bench.py --config ${CONFIG:-3} 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('  build_ms median %.4f min %.4f' % (d['build_ms_median'], d['build_ms_min']))"
  if [ -n "$TRACE" ]; then
    rm -rf /tmp/abst && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abst -- python3 tools/build_bench.py --config ${CONFIG:-3} --reps 10 > /dev/null 2>&1
    python3 tools/kstats.py /tmp/abst | grep -E "$TRACE"
  fi
done
rm -f binary_amd/csrc/build.o && make -C binary_amd/csrc -s
