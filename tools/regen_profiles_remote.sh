#!/bin/bash
# The GPU-box half of tools/regen_profiles.sh: every measurement the round's profiles/ files come from, from the tree as it
# is, into gpurun_out/regen/. BIVX_GIT_REV (set by the caller) is stamped into every bench line. Each step writes its own
# files, so a step that fails leaves the others' results behind. usage (through gpurun): bash tools/regen_profiles_remote.sh [part ...]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; export TMPDIR=/tmp
O=$R/gpurun_out/regen; mkdir -p "$O"
PARTS=${*:-bench profile3 profile2 profile5 sorted issue build matrix misc rehearse}
say() { echo "[regen $(date +%H:%M:%S)] $*"; }
for part in $PARTS; do
  say "part $part"
  case $part in
    bench)
      timeout -k 10 600 python3 bench.py > "$O/bench_default_with_cpu_baseline.json" 2> "$O/bench_default.err"
      timeout -k 10 300 python3 bench.py --config 2 > "$O/bench_config2.json" 2> "$O/bench_config2.err"
      timeout -k 10 900 python3 bench.py --config 5 --steps 10 --warmup 3 > "$O/bench_config5.json" 2> "$O/bench_config5.err" ;;
    profile3) timeout -k 10 900 bash tools/profile_bench.sh regen 3 > "$O/profile3.log" 2>&1 ;;
    profile2) timeout -k 10 600 bash tools/profile_bench.sh regen 2 > "$O/profile2.log" 2>&1 ;;
    profile5) timeout -k 10 1000 bash tools/profile_bench.sh regen 5 > "$O/profile5.log" 2>&1 ;;
    sorted)
      timeout -k 10 900 bash tools/profile_bench.sh regen_sorted 3 --sorted-queries > "$O/profile3s.log" 2>&1
      timeout -k 10 1000 bash tools/profile_bench.sh regen_sorted 5 --sorted-queries > "$O/profile5s.log" 2>&1 ;;
    issue)
      timeout -k 10 600 bash tools/pmc_issue.sh regen 3 > "$O/issue3.log" 2>&1
      timeout -k 10 600 bash tools/pmc_issue.sh regen_sorted 3 --sorted-queries > "$O/issue3s.log" 2>&1
      timeout -k 10 900 bash tools/pmc_issue.sh regen_sorted 5 --sorted-queries > "$O/issue5s.log" 2>&1 ;;
    build) timeout -k 10 600 bash tools/profile_build.sh regen 3 > "$O/build3.log" 2>&1 ;;
    matrix) timeout -k 10 900 bash tools/perf_matrix.sh > "$O/perf_matrix.txt" 2>&1 ;;
    misc)
      timeout -k 10 300 python3 tools/shard_sizes.py > "$O/shard_sizes.txt" 2>&1
      gcc -std=c11 -O2 -I include tests/c/concurrent_queries.c -o /tmp/cq -L binary_amd -lbivx -pthread -Wl,-rpath,$R/binary_amd &&
        timeout -k 10 300 /tmp/cq 8 2000 > "$O/concurrent_queries.json" 2>&1
      timeout -k 10 300 python3 tools/two_process_stress.py 1e7 80 > "$O/two_process_stress.txt" 2>&1
      timeout -k 10 300 python3 tools/create_build_drop.py > "$O/create_build_drop.txt" 2>&1 ;;
    rehearse)
      export BIVX_BENCH_DEVICE=0 BIVX_BENCH_BACKEND=gloo
      timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29521 \
        bench.py --gpus 2 --steps 20 --warmup 3 2> "$O/rehearse4.err" | grep '^{' > "$O/bench_config4_2ranks_one_gpu_gloo_rehearsal.json"
      timeout -k 10 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29522 \
        bench.py --gpus 2 --config 5 --steps 5 --warmup 2 2> "$O/rehearse5.err" | grep '^{' > "$O/bench_config5_2ranks_one_gpu_gloo_rehearsal.json"
      unset BIVX_BENCH_DEVICE BIVX_BENCH_BACKEND ;;
  esac
done
say "done"; ls -la "$O" | tail -40
