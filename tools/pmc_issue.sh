#!/bin/bash
# Issue-side counters of the dominant query kernel under `bench.py --config C`: instructions by class, busy / wait
# cycles, LDS bank conflicts — separate rocprofv3 --pmc passes (never combined with a trace), averaged per launch by
# tools/pmc_issue.py. usage (through gpurun):  bash tools/pmc_issue.sh <tag> [config=3] [extra bench flags...]
set -eo pipefail
TAG=${1:-run}
CFG=${2:-3}
shift || true
shift || true
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/issue_${TAG}_c$CFG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
BP="python3 bench.py --config $CFG --steps 6 --warmup 2 --no-cpu-baseline --no-parity --no-extras $*"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT" \
           "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_INST_CYCLES_SALU" \
           "GRBM_GUI_ACTIVE SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc/p$i" -- $BP > "$OUT/p$i.log" 2>&1 || echo "pass $i ($set) failed" >> "$OUT/failed.txt"
done
python3 tools/pmc_issue.py "$OUT/pmc" > "$R/gpurun_out/pmc_issue_config${CFG}_$TAG.txt"
rm -rf "$OUT/pmc"
cat "$R/gpurun_out/pmc_issue_config${CFG}_$TAG.txt"
