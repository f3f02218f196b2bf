#!/bin/bash
# Issue-side counters of ONE kernel under any python command: separate rocprofv3 --pmc passes (never combined with a
# trace), averaged per launch over the dispatches whose kernel name contains KERNEL.
# usage (through gpurun):  bash tools/pmc_any.sh <tag> <kernel substring> <python script> [args...]
set -eo pipefail
TAG=$1; KERNEL=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_any_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT" \
           "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_INST_CYCLES_SALU" \
           "GRBM_GUI_ACTIVE SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d "$OUT/p$i" -- python3 "$@" > "$OUT/p$i.log" 2>&1 || echo "pass $i ($set) failed" >> "$OUT/failed.txt"
done
python3 - "$OUT" "$KERNEL" > "$R/gpurun_out/pmc_any_$TAG.txt" <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"kernel *{sys.argv[2]}*; averages per launch over {max((len(v) for v in acc.values()), default=0)} launches")
for k in sorted(acc):
    v = acc[k]
    print(f"  {k:28s} {sum(v) / len(v):16.0f}")
PY
rm -rf "$OUT"/p[0-9]*
cat "$R/gpurun_out/pmc_any_$TAG.txt"
