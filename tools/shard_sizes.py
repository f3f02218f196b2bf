#!/usr/bin/env python3
"""What a rank of `bench.py --gpus N` runs (config 4: config 3's set, chromosomes LPT-assigned), timed on ONE GPU for
N = 2, 4, 8 with the pipelined kernel forced (BIVX_PIPE=2) and off (BIVX_PIPE=0): where the batch-size threshold of
pipe_eligible belongs. Diagnostic; prints one line per (N, kernel)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binary_amd import IntervalIndex, synth, sharding  # noqa: E402

dev = torch.device("cuda:0")
to = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)
ni, nq = synth.split_by_length(10_000_000), synth.split_by_length(10_000_000)
for world in (2, 4, 8):
    chroms = sharding.lpt_assign(sharding.chrom_work(ni, nq), world)[0]
    d = synth.gen_genome(10_000_000, 10_000_000, 1000, chrom_ids=chroms)
    idx = IntervalIndex(0)
    idx.insert_node(to(d["low"]), to(d["high"]), to(d["chrom"]))
    idx.build()
    qc, ql, qh = to(d["qchrom"]), to(d["qlow"]), to(d["qhigh"])
    Q = ql.numel()
    off = torch.empty(Q + 1, dtype=torch.int64, device=dev)
    idx.count_overlaps_device(ql, qh, qc, offsets=off)
    H = int(off[-1].item())
    hits = torch.empty(H, dtype=torch.int32, device=dev)
    for mode in ("0", "2"):
        os.environ["BIVX_PIPE"] = mode
        for _ in range(5):
            idx.query_device(ql, qh, off, hits, qchrom=qc)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            idx.query_device(ql, qh, off, hits, qchrom=qc)
        e1.record()
        torch.cuda.synchronize()
        print(f"N={world} rank 0: {Q} queries, {H} ids, BIVX_PIPE={mode}: {e0.elapsed_time(e1) / 50 * 1000:.1f} us", flush=True)
    idx.close()
