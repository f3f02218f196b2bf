#!/usr/bin/env python3
"""Times the two-pass count + fill on positionally clustered intervals (hotspots), where directory windows are long."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from binary_amd import IntervalIndex
rng = np.random.default_rng(33)
low = np.concatenate([rng.integers(5_000_000, 5_020_000, size=300_000), rng.integers(90_000_000, 90_000_400, size=50_000),
                      rng.integers(0, 200_000_000, size=650_000)]).astype(np.uint32)
high = low + rng.integers(0, 300, size=low.size).astype(np.uint32)
qlo = np.concatenate([rng.integers(4_999_000, 5_021_000, size=100_000), rng.integers(89_999_900, 90_000_500, size=20_000),
                      rng.integers(0, 200_000_000, size=880_000)]).astype(np.uint32)
rng.shuffle(qlo)
qhi = qlo + rng.integers(0, 50, size=qlo.size).astype(np.uint32)
dev = torch.device("cuda:0")
to = lambda a: torch.from_numpy(a.view(np.int32)).to(dev)
idx = IntervalIndex(0); idx.insert_node(low, high); idx.build()
dql, dqh = to(qlo), to(qhi)
off = torch.empty(qlo.size + 1, dtype=torch.int64, device=dev)
idx.count_overlaps_device(dql, dqh, offsets=off)
H = int(off[-1].item())
hits = torch.empty(H, dtype=torch.int32, device=dev)
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
print(f"H={H}  single-pass {timed(lambda: idx.query_device(dql, dqh, off, hits)):.3f} ms   count {timed(lambda: idx.count_overlaps_device(dql, dqh, offsets=off)):.3f} ms")
