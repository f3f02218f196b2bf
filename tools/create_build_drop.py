#!/usr/bin/env python3
"""The reference's pattern — a tree per task: created, filled, built, queried once, dropped (mapper.hpp:147-162,199) — timed
per stage on device-resident columns: bivx_create, bivx_append_dev, bivx_build, one query batch, bivx_destroy.
usage: create_build_drop.py [intervals=1000000] [rounds=8] [--host: host arrays in, host CSR out]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from binary_amd import IntervalIndex, synth  # noqa: E402

HOST = "--host" in sys.argv
args = [a for a in sys.argv[1:] if a != "--host"]
n = int(float(args[0])) if len(args) > 0 else 1_000_000
rounds = int(args[1]) if len(args) > 1 else 8
dev = torch.device("cuda:0")
lo, hi = synth.gen_intervals(n, 248_956_422, 1000, 0)
ql, qh = synth.gen_point_queries(min(n, 1_000_000), 248_956_422, 0)
to = lambda a: torch.from_numpy(a.view(np.int32)).to(dev)
d_lo, d_hi, d_ql, d_qh = to(lo), to(hi), to(ql), to(qh)
off = torch.empty(ql.size + 1, dtype=torch.int64, device=dev)
hits = torch.empty(8 * ql.size, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
rows = []
for r in range(rounds):
    t = [time.perf_counter()]
    idx = IntervalIndex(0); t.append(time.perf_counter())
    if HOST:
        idx.insert_node(lo, hi); t.append(time.perf_counter())
        idx.build(); t.append(time.perf_counter())
        idx.find_overlaps(ql, qh, sort_by_id=False); t.append(time.perf_counter())
    else:
        idx.insert_node(d_lo, d_hi); torch.cuda.synchronize(); t.append(time.perf_counter())
        idx.build(); t.append(time.perf_counter())
        idx.query_device(d_ql, d_qh, off, hits); torch.cuda.synchronize(); t.append(time.perf_counter())
    idx.close(); t.append(time.perf_counter())
    rows.append(np.diff(t) * 1e3)
rows = np.array(rows)
names = ["create", "append" if HOST else "append_dev", "build", "query", "destroy"]
print(f"n={n}: ms per stage, first round then median of the rest")
for k, nm in enumerate(names):
    print(f"  {nm:10s} {rows[0, k]:8.3f}  {np.median(rows[1:, k]):8.3f}")
print(f"  {'total':10s} {rows[0].sum():8.3f}  {np.median(rows[1:].sum(axis=1)):8.3f}")
