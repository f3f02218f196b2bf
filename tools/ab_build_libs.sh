#!/bin/bash
# A/B of prebuilt library variants (tools/build_variant.sh <tag> "<flags>" build) on the build: tools/build_bench.py times and the
# build kernels' averages from one traced run per variant. usage: [CONFIG=3] [TRACE='bin_stats|scatter'] tools/ab_build_libs.sh tag...
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; export TMPDIR=/tmp
for tag in "$@"; do
  if [ "$tag" = default ]; then unset BIVX_LIB; else export BIVX_LIB=$R/binary_amd/libbivx.so.$tag; fi
  echo "== $tag"
  timeout -k 10 200 python3 tools/build_bench.py --config ${CONFIG:-3} 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('  build_ms %.4f  append_ms %.4f  append+build %.4f' % (d['build_ms_median'], d['append_ms_median'], d['append_plus_build_ms_median']), flush=True)"
  if [ -n "$TRACE" ]; then
    rm -rf /tmp/abst && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abst -- python3 tools/build_bench.py --config ${CONFIG:-3} --reps 10 > /dev/null 2>&1
    python3 tools/kstats.py /tmp/abst | grep -E "$TRACE"
  fi
done
