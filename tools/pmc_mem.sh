#!/bin/bash
# Texture-addresser busy cycles of the dominant query kernel under `bench.py --config C`, one rocprofv3 --pmc pass
# averaged per launch by tools/pmc_issue.py.
# usage (through gpurun):  bash tools/pmc_mem.sh <tag> [config=3] [extra bench flags...]
set -eo pipefail
TAG=${1:-run}
CFG=${2:-3}
shift || true
shift || true
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/mem_${TAG}_c$CFG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
BP="python3 bench.py --config $CFG --steps 6 --warmup 2 --no-cpu-baseline --no-parity --no-extras $*"
i=0
# (only this set: a pass with TA_ADDR_STALLED_BY_TC_CYCLES_sum / TA_DATA_STALLED_BY_TC_CYCLES_sum / TA_FLAT_*_WAVEFRONTS_sum
#  aborted rocprofv3 and hung the run on this pool — do not add them back)
for set in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr TA_BUSY_max"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc/p$i" -- $BP > "$OUT/p$i.log" 2>&1 || echo "pass $i ($set) failed" >> "$OUT/failed.txt"
done
python3 tools/pmc_issue.py "$OUT/pmc" > "$R/gpurun_out/pmc_mem_config${CFG}_$TAG.txt"
rm -rf "$OUT/pmc"
cat "$R/gpurun_out/pmc_mem_config${CFG}_$TAG.txt"
[ -f "$OUT/failed.txt" ] && cat "$OUT/failed.txt" || true
