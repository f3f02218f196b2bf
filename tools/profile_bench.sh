#!/bin/bash
# Profiles the default `bench.py` command on the GPU box: rocprofv3 kernel trace + stats, then separate --pmc
# passes for HBM-side traffic (MI355X_MICROARCH.md, HBM section). Writes under gpurun_out/prof_<tag>/ and
# gpurun_out/pmc_traffic_<tag>.json; copy what is to be judged into profiles/.
# usage (through gpurun):  bash tools/profile_bench.sh <tag>
set -eo pipefail
TAG=${1:-run}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
B="python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline"
$B > "$OUT/bench.json" 2> "$OUT/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $B --no-phases > "$OUT/stats.log" 2>&1
python3 tools/kstats.py "$OUT/stats" > "$OUT/kernel_stats.txt"
BP="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-phases"
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d "$OUT/pmc/a" -- $BP > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc/b" -- $BP > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc/c" -- $BP > /dev/null 2>&1
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc/d" -- $BP > /dev/null 2>&1
python3 tools/pmc_traffic.py "$OUT/pmc" "$R/gpurun_out/pmc_traffic_$TAG.json"
find "$OUT/stats" -name "*kernel_stats.csv" -exec cp {} "$R/gpurun_out/kernel_stats_$TAG.csv" \;
rm -rf "$OUT/pmc" "$OUT/stats"   # raw per-dispatch CSVs are large; the summaries above are what is kept
echo "profile $TAG done"
