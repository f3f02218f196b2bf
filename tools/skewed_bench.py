#!/usr/bin/env python3
"""SV-like length spectrum: 1 M intervals with log-uniform lengths 50 bp .. MAXLEN (argv[1], default 10 Mbp) on one
chromosome, 1 M point queries (argv[2]: another number of queries). Prints the length classes the planner chose and the single-pass time."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from binary_amd import IntervalIndex
rng = np.random.default_rng(5)
L = 248_956_422
n = 1_000_000
q = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
low = rng.integers(0, L - 10_000_001, size=n).astype(np.uint32)
MAXLEN = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
ln = np.exp(rng.uniform(np.log(50), np.log(MAXLEN), size=n)).astype(np.uint32)
high = low + ln
qlo = rng.integers(0, L, size=q).astype(np.uint32)
dev = torch.device("cuda:0")
to = lambda a: torch.from_numpy(a.view(np.int32)).to(dev)
idx = IntervalIndex(0); idx.insert_node(low, high); idx.build()
st = idx.stats()
dq = to(qlo)
off = torch.empty(q + 1, dtype=torch.int64, device=dev)
idx.count_overlaps_device(dq, dq, offsets=off)
H = int(off[-1].item())
hits = torch.empty(H, dtype=torch.int32, device=dev)
def timed(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
t = timed(lambda: idx.query_device(dq, dq, off, hits))
t2 = timed(lambda: idx.count_overlaps_device(dq, dq, offsets=off))
print(f"maxlen={MAXLEN} count-only {t2:.3f} ms;", end=" ")
print(f"segments={st['n_segments']} build_ms={st['build_ms']:.2f} H={H} ({H/q:.1f} hits/query) single-pass {t:.3f} ms = {q/t/1e6:.2f} G q/s, {H/t/1e6:.2f} G hits/s")
