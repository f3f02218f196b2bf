#!/bin/bash
# One table of every kernel variant's time on its reference workload (run on the GPU box, ~2 GPU-minutes): the
# numbers to compare against profiles/*perf_matrix.txt before and after touching shared device code — the variants
# live within 64 VGPRs and react to each other's changes.
set -e
cd "$(dirname "$0")/.."
echo "== config 2, 1 M x 1 M random points (ordered / unordered / count): lean kernel"
python tools/quick_modes.py 2>&1 | grep ordered
echo "== configs 2-3: (single pass, with ordered ids [S variants], unordered [U variants]) ms"
python tools/measure_configs.py 2>/dev/null | python -c "
import json,sys
s=sys.stdin.read(); d=json.loads(s[s.index('{'):])
for c in d['results']:
    print('  ', c['name'], {o:(round(c[o]['single_pass_ms'],4), round(c[o]['single_pass_sorted_ids_ms'],4), round(c[o]['unordered_begin_count_ms'],4)) for o in ('random','sorted')}, 'host api ms', round(c['host_api_ms'],2))"
echo "== SV-like length spectra, several segments per query [MS variants]"
for m in 1e4 1e5 1e6 1e7; do python tools/skewed_bench.py $m 2>&1 | grep maxlen | cut -c1-200; done
echo "== positional hotspots (wavefront-cooperative path, trimming)"
python tools/clustered_bench.py 2>&1 | grep single-pass
echo "== one query per call through the host entry points"
python tools/single_query_latency.py 2>&1 | grep bivx_
