#!/bin/bash
# Rebuilds libbivx.so with different single-pass tile shapes and times bench.py for each (run on the GPU box).
set -e
cd "$(dirname "$0")/.."
for cfg in "1024" "512" "256"; do
  set -- $cfg
  make -C binary_amd/csrc -s clean
  make -C binary_amd/csrc -s -j8 EXTRA="-DBIVX_FUSED_THREADS=$1" 2>&1 | grep -E "error" || true
  echo "== threads per tile=$1"
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline $BENCH_EXTRA | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   ms/step', round(d['ms_per_step'],4), 'Gq/s', round(d['value']/1e9,2), d['parity'])"
done
make -C binary_amd/csrc -s clean
make -C binary_amd/csrc -s -j8
