#!/usr/bin/env python3
"""Config 3 with a position-sorted batch in which a fraction of the queries is displaced (merged or lightly shuffled
inputs): how much of the streaming path's gain survives. Diagnostic."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binary_amd import IntervalIndex, synth  # noqa: E402

dev = torch.device("cuda:0")
to = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)
d = synth.gen_genome(10_000_000, 10_000_000, 1000)
idx = IntervalIndex(0)
idx.insert_node(to(d["low"]), to(d["high"]), to(d["chrom"]))
idx.build()
p0 = np.lexsort((d["qlow"], d["qchrom"]))
rng = np.random.default_rng(0)
Q = p0.size
off = torch.empty(Q + 1, dtype=torch.int64, device=dev)
hits = torch.empty(40_000_000, dtype=torch.int32, device=dev)
for frac in (0.0, 0.001, 0.01, 0.05, 0.2, 1.0):
    p = p0.copy()
    k = int(Q * frac)
    if k:
        sw = rng.permutation(Q)[:k]
        p[sw] = p[np.roll(sw, 1)]
    qc, ql, qh = to(d["qchrom"][p]), to(d["qlow"][p]), to(d["qhigh"][p])
    for _ in range(3):
        idx.query_device(ql, qh, off, hits, qchrom=qc)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        idx.query_device(ql, qh, off, hits, qchrom=qc)
    e1.record()
    torch.cuda.synchronize()
    print(f"{frac * 100:5.1f} % of the queries displaced: {e0.elapsed_time(e1) / 20:.4f} ms", flush=True)
