#!/usr/bin/env python3
"""Config 2 (1 M x 1 M random point queries): device time of the ordered and the unordered single pass."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binary_amd import IntervalIndex, synth  # noqa: E402

dev = torch.device("cuda:0")
L = int(synth.HG38_LENGTHS[0])
lo, hi = synth.gen_intervals(1_000_000, L, 1000, 0)
ql, qh = synth.gen_point_queries(1_000_000, L, 0)
to = lambda a: torch.from_numpy(a.view(np.int32)).to(dev)
idx = IntervalIndex(0)
idx.insert_node(lo, hi)
idx.build()
Q = ql.size
dql, dqh = to(ql), to(qh)
off = torch.empty(Q + 1, dtype=torch.int64, device=dev)
hits = torch.empty(3_000_000, dtype=torch.int32, device=dev)
beg = torch.empty(Q, dtype=torch.int64, device=dev)
cnt = torch.empty(Q, dtype=torch.int32, device=dev)
tot = torch.zeros(1, dtype=torch.int64, device=dev)


def timed(fn, reps=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


a = timed(lambda: idx.query_device(dql, dqh, off, hits))
H = int(off[-1].item())
b = timed(lambda: idx.query_device_unordered(dql, dqh, beg, cnt, hits, tot))
assert int(tot.item()) == H and int(cnt.sum().item()) == H
c = timed(lambda: idx.count_overlaps_device(dql, dqh, offsets=off))
print(f"ordered {a:.1f} us   unordered {b:.1f} us   count-only {c:.1f} us   (H = {H})")
