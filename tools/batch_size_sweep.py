#!/usr/bin/env python3
"""Where the batch-size threshold of pipe_eligible belongs: config 2's index (1 M intervals on chr1), point queries in
generation order, batches of 0.125 M .. 2 M queries through k_query_fused (BIVX_PIPE=0) and through k_query_pipe
(BIVX_PIPE=2). Diagnostic; one line per size."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binary_amd import IntervalIndex, synth  # noqa: E402

dev = torch.device("cuda:0")
to = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)
L = int(synth.HG38_LENGTHS[0])
low, high = synth.gen_intervals(1_000_000, L, 1000, 0)
idx = IntervalIndex(0)
idx.insert_node(to(low), to(high))
idx.build()
SIZES = [int(x) for x in sys.argv[1:]] or [131072, 262144, 393216, 524288, 655360, 786432, 1000000, 1310720, 2097152]
for q in SIZES:
    qlo, qhi = synth.gen_point_queries(q, L, 0)
    ql, qh = to(qlo), to(qhi)
    off = torch.empty(q + 1, dtype=torch.int64, device=dev)
    idx.count_overlaps_device(ql, qh, offsets=off)
    H = int(off[-1].item())
    hits = torch.empty(max(H, 1), dtype=torch.int32, device=dev)
    res = {}
    for mode in ("0", "2"):
        os.environ["BIVX_PIPE"] = mode
        for _ in range(10):
            idx.query_device(ql, qh, off, hits)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            idx.query_device(ql, qh, off, hits)
        e1.record()
        torch.cuda.synchronize()
        res[mode] = e0.elapsed_time(e1) / 200 * 1000
    print(f"{q:8d} queries: k_query_fused {res['0']:6.1f} us   k_query_pipe {res['2']:6.1f} us", flush=True)
