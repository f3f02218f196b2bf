#!/bin/bash
# A/B of prebuilt library variants (tools/build_variant.sh) on the GPU box: times of config C in both query orders, then
# (PMC=1) the instruction counters of the full call per order. usage: [PMC=1] [KERNEL=k_query_pipe] tools/ab_libs.sh C tag...
# ("default" = binary_amd/libbivx.so)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; export TMPDIR=/tmp
C=$1; shift
KERNEL=${KERNEL:-k_query_pipe}
for tag in "$@"; do
  if [ "$tag" = default ]; then unset BIVX_LIB; else export BIVX_LIB=$R/binary_amd/libbivx.so.$tag; fi
  timeout -k 10 300 python3 tools/ab_one.py $C $ABFLAGS 2>/dev/null | grep '^{' | python3 -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l)
    print('$tag', ' | '.join(f\"{o}: full {d[o]['full_ms']:.4f} count {d[o]['count_ms']:.4f} cks {d[o]['checksum']}\" for o in ('gen','sorted') if o in d), flush=True)"
  if [ -n "$PMC" ]; then
    for only in ${ONLY:-gen:full sorted:full}; do
      O=/tmp/abpmc_$$; rm -rf $O
      rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O -- python3 tools/ab_one.py $C --only $only $ABFLAGS > /dev/null 2>&1
      rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/b -- python3 tools/ab_one.py $C --only $only $ABFLAGS > /dev/null 2>&1
      python3 - $O "$KERNEL" "$tag $only" <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"] and "fill" not in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
w = m.get("SQ_WAVES", 0)
# slices of 64 queries: fifteen of every sixteen wavefronts are workers, each slice is one worker iteration
print(f"  pmc {sys.argv[3]}: " + "  ".join(f"{k[3:]} {v/1e6:.2f}M" for k, v in sorted(m.items())), flush=True)
PY
      rm -rf $O
    done
  fi
done
