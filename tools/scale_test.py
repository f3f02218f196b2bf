#!/usr/bin/env python3
"""BASELINE.json configs[4] on ONE GPU: whole-genome self-overlap (queries = the intervals), N = Q = 50 M over
24 chromosomes with counts proportional to length. Checks the size-independent properties (per-query count equals
the sort+searchsorted count; a strided sample of hit lists satisfies the predicate and ascends) and prints timings."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from binary_amd import IntervalIndex, synth  # noqa: E402
from oracle import ivtree_oracle as oracle  # noqa: E402  (checker only)

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
dev = torch.device("cuda:0")
t0 = time.perf_counter()
data = synth.gen_genome(N, 0, 1000)
gen_s = time.perf_counter() - t0
to = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)
d_c, d_lo, d_hi = to(data["chrom"]), to(data["low"]), to(data["high"])
idx = IntervalIndex(0)
t0 = time.perf_counter()
idx.insert_node(d_lo, d_hi, d_c)
idx.build()
build_s = time.perf_counter() - t0
st = idx.stats()
off = torch.empty(N + 1, dtype=torch.int64, device=dev)
ws = torch.empty(idx.count_workspace_bytes(N), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
idx.count_overlaps_device(d_lo, d_hi, d_c, offsets=off, workspace=ws)
H = int(off[-1].item())
count_s = time.perf_counter() - t0
hits = torch.empty(H, dtype=torch.int32, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
idx.query_device(d_lo, d_hi, off, hits, qchrom=d_c, sort_by_id=False)
e1.record()
torch.cuda.synchronize()
single_ms = e0.elapsed_time(e1)
e0.record()
idx.query_device(d_lo, d_hi, off, hits, qchrom=d_c, sort_by_id=True)
e1.record()
torch.cuda.synchronize()
sorted_ms = e0.elapsed_time(e1)
off_h = off.cpu().numpy()
cnt = np.diff(off_h)
ok_counts = True
for c in range(24):
    sel = data["chrom"] == c
    exp = oracle.count_overlaps_numpy(data["low"][sel], data["high"][sel], data["low"][sel], data["high"][sel])
    ok_counts &= bool(np.array_equal(cnt[sel], exp))
# strided sample of queries: predicate + ascending + same chromosome
samp = np.arange(0, N, max(1, N // 200_000))
ok_hits = True
hits_h = hits.cpu().numpy().view(np.uint32)
for q in samp[:20000]:
    h = hits_h[off_h[q]:off_h[q + 1]]
    ok_hits &= bool(np.all(data["chrom"][h] == data["chrom"][q]) and np.all(data["low"][h] <= data["high"][q])
                    and np.all(data["high"][h] >= data["low"][q]) and np.all(np.diff(h.astype(np.int64)) > 0) and (q in h))
print(json.dumps({"N": N, "H": H, "gen_s": gen_s, "append+build_s": build_s, "build_ms": st["build_ms"],
                  "segments": st["n_segments"], "index_bytes": st["index_bytes"], "count_two_pass_s": count_s,
                  "single_pass_ms": single_ms, "single_pass_sorted_ids_ms": sorted_ms,
                  "gqps_single_pass": N / single_ms / 1e6, "counts_exact": ok_counts, "sampled_hit_lists_ok": ok_hits}))
