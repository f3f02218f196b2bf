#!/bin/bash
# Profiles `bench.py --config C` on the GPU box: rocprofv3 kernel trace + stats, then separate --pmc passes for
# the fabric/HBM-side traffic (MI355X_MICROARCH.md, HBM section: FETCH_SIZE tallies 128-B requests at 64 B, so the
# read side is taken from the TCC_EA0_RDREQ request-size split). Writes gpurun_out/prof_<tag>/, and
# gpurun_out/pmc_traffic_config<C>_<tag>.json + gpurun_out/kernel_stats_config<C>_<tag>.csv; copy what is to be judged
# into profiles/ (pmc_traffic_config<C>.json is the file bench.py attaches to its line).
# usage (through gpurun):  bash tools/profile_bench.sh <tag> [config=3] [extra bench flags...]
set -eo pipefail
TAG=${1:-run}
CFG=${2:-3}
shift || true
shift || true
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_${TAG}_c$CFG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$R"
case $CFG in
  2) STEPS=200; PSTEPS=20 ;;
  3) STEPS=50;  PSTEPS=10 ;;
  *) STEPS=5;   PSTEPS=3 ;;
esac
B="python3 bench.py --config $CFG --steps $STEPS --warmup 5 --no-cpu-baseline --no-parity --no-extras $*"
$B > "$OUT/bench.json" 2> "$OUT/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $B > "$OUT/stats.log" 2>&1
python3 tools/kstats.py "$OUT/stats" > "$OUT/kernel_stats.txt"
BP="python3 bench.py --config $CFG --steps $PSTEPS --warmup 2 --no-cpu-baseline --no-parity --no-extras $*"
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d "$OUT/pmc/a" -- $BP > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc/b" -- $BP > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc/c" -- $BP > /dev/null 2>&1
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc/d" -- $BP > /dev/null 2>&1
python3 tools/pmc_traffic.py "$OUT/pmc" "$R/gpurun_out/pmc_traffic_config${CFG}_$TAG.json" "$OUT/bench.json" > "$OUT/pmc_traffic.txt"
find "$OUT/stats" -name "*kernel_stats.csv" -exec cp {} "$R/gpurun_out/kernel_stats_config${CFG}_$TAG.csv" \;
rm -rf "$OUT/pmc" "$OUT/stats"   # raw per-dispatch CSVs are large; the summaries above are what is kept
echo "profile $TAG config $CFG done"
