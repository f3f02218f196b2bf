#!/bin/bash
# Profiles bivx_build on the GPU box (tools/build_bench.py): wall time, rocprofv3 kernel trace + stats, then separate --pmc
# passes for the build kernels' memory-side traffic. Writes gpurun_out/build_<tag>_c<C>.{json,kernel_stats.txt,pmc.json}.
# usage (through gpurun):  bash tools/profile_build.sh <tag> [config=3] [extra build_bench flags...]
set -eo pipefail
TAG=${1:-run}
CFG=${2:-3}
shift || true
shift || true
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/build_${TAG}_c$CFG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
python3 tools/build_bench.py --config $CFG $* > "$OUT.json" 2> "$OUT/err.txt"
N=$(python3 -c "import json;print(json.load(open('$OUT.json'))['intervals'])")
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 tools/build_bench.py --config $CFG --reps 10 $* > "$OUT/stats.log" 2>&1
python3 tools/kstats.py "$OUT/stats" > "$OUT.kernel_stats.txt"
B="python3 tools/build_bench.py --config $CFG --reps 4 $*"
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d "$OUT/pmc/a" -- $B > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc/b" -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc/c" -- $B > /dev/null 2>&1
python3 tools/pmc_build.py "$OUT/pmc" $N > "$OUT.pmc.json"
rm -rf "$OUT/pmc" "$OUT/stats"
cat "$OUT.json" "$OUT.kernel_stats.txt"
echo "build profile $TAG config $CFG done"
