#!/bin/bash
# Regenerates the round's evidence from HEAD: runs tools/regen_profiles_remote.sh on a GPU box (gpurun), then copies the
# summaries into profiles/ under the round's prefix, every JSON line stamped with `git rev-parse HEAD` (bench.py's "commit",
# through BIVX_GIT_REV: the box has no .git) and every text file headed by it. A dirty tree is refused: the stamp would lie.
# usage: tools/regen_profiles.sh [round prefix = r04] [part ...]     parts: see regen_profiles_remote.sh
set -eo pipefail
cd "$(dirname "$0")/.."
P=${1:-r04}; shift || true
if [ -n "$(git status --porcelain --untracked-files=no)" ]; then echo "commit first: the tree is dirty"; exit 1; fi
REV=$(git rev-parse HEAD)
python -c "import __graft_entry__ as g; g.build()"
MARK=$(mktemp); touch "$MARK"   # only what THIS run wrote is copied (a partial run must not restamp the other parts' files)
/usr/local/graft/bin/gpurun --timeout 1200 -- "BIVX_GIT_REV=$REV bash tools/regen_profiles_remote.sh $*"
O=gpurun_out/regen
stamp_txt() { [ -s "$1" ] && [ "$1" -nt "$MARK" ] && { echo "# commit $REV ($(date -u +%Y-%m-%dT%H:%MZ)); tools/regen_profiles.sh"; grep -v 'amdgpu.ids' "$1"; } > "$2" || true; }
cp_json() { [ -s "$1" ] && [ "$1" -nt "$MARK" ] && grep '^{' "$1" | tail -1 > "$2" || true; }
cp_json $O/bench_default_with_cpu_baseline.json profiles/${P}_bench_default_with_cpu_baseline.json
cp_json $O/bench_config2.json profiles/${P}_bench_config2.json
cp_json $O/bench_config5.json profiles/${P}_bench_config5.json
cp_json $O/bench_config4_2ranks_one_gpu_gloo_rehearsal.json profiles/${P}_bench_config4_2ranks_one_gpu_gloo_rehearsal.json
cp_json $O/bench_config5_2ranks_one_gpu_gloo_rehearsal.json profiles/${P}_bench_config5_2ranks_one_gpu_gloo_rehearsal.json
cp_json $O/concurrent_queries.json profiles/${P}_concurrent_queries.json
for c in 2 3 5; do
  fresh() { [ -s "$1" ] && [ "$1" -nt "$MARK" ]; }
  fresh gpurun_out/pmc_traffic_config${c}_regen.json && { cp gpurun_out/pmc_traffic_config${c}_regen.json profiles/${P}_pmc_traffic_config$c.json; cp gpurun_out/pmc_traffic_config${c}_regen.json profiles/pmc_traffic_config$c.json; }
  fresh gpurun_out/kernel_stats_config${c}_regen.csv && cp gpurun_out/kernel_stats_config${c}_regen.csv profiles/${P}_kernel_stats_config$c.csv
  fresh gpurun_out/pmc_traffic_config${c}_regen_sorted.json && cp gpurun_out/pmc_traffic_config${c}_regen_sorted.json profiles/${P}_pmc_traffic_config${c}_position_sorted.json
  fresh gpurun_out/kernel_stats_config${c}_regen_sorted.csv && cp gpurun_out/kernel_stats_config${c}_regen_sorted.csv profiles/${P}_kernel_stats_config${c}_position_sorted.csv
  fresh gpurun_out/pmc_issue_config${c}_regen.txt && stamp_txt gpurun_out/pmc_issue_config${c}_regen.txt profiles/${P}_issue_counters_config${c}_generation_order.txt
  fresh gpurun_out/pmc_issue_config${c}_regen_sorted.txt && stamp_txt gpurun_out/pmc_issue_config${c}_regen_sorted.txt profiles/${P}_issue_counters_config${c}_position_sorted.txt
done
[ -s gpurun_out/build_regen_c3.json ] && cp_json gpurun_out/build_regen_c3.json profiles/${P}_build_config3.json
[ -s gpurun_out/build_regen_c3.kernel_stats.txt ] && stamp_txt gpurun_out/build_regen_c3.kernel_stats.txt profiles/${P}_build_kernel_stats_config3.txt
[ -s gpurun_out/build_regen_c3.pmc.json ] && [ gpurun_out/build_regen_c3.pmc.json -nt "$MARK" ] && cp gpurun_out/build_regen_c3.pmc.json profiles/${P}_build_pmc_traffic_config3.json
stamp_txt $O/perf_matrix.txt profiles/${P}_perf_matrix.txt
stamp_txt $O/shard_sizes.txt profiles/${P}_shard_sizes.txt
stamp_txt $O/two_process_stress.txt profiles/${P}_two_process_stress.txt
stamp_txt $O/create_build_drop.txt profiles/${P}_create_build_drop.txt
echo "profiles/${P}_* regenerated at $REV"
