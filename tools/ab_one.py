#!/usr/bin/env python3
"""One library variant (BIVX_LIB=...), one BASELINE config: device time of the single-pass call in generation order and
position-sorted, full CSR and offsets only. With --only ORDER:MODE just that call a few times (what tools/ab_libs.sh
profiles with rocprofv3 --pmc). Prints one JSON line. Diagnostic."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binary_amd import IntervalIndex, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("config", type=int)
ap.add_argument("--only", default=None)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--sort-ids", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
to = lambda x: torch.from_numpy(np.ascontiguousarray(x).view(np.int32)).to(dev)
if a.config == 2:
    L = int(synth.HG38_LENGTHS[0])
    lo, hi = synth.gen_intervals(1_000_000, L, 1000, 0)
    ql, qh = synth.gen_point_queries(1_000_000, L, 0)
    z = np.zeros(1_000_000, np.uint32)
    d = dict(chrom=z, low=lo, high=hi, qchrom=z, qlow=ql, qhigh=qh)
elif a.config == 3:
    d = synth.gen_genome(10_000_000, 10_000_000, 1000)
else:
    d = synth.gen_genome(50_000_000, 0, 1000)
    d.update(qchrom=d["chrom"], qlow=d["low"], qhigh=d["high"])
idx = IntervalIndex(0)
idx.insert_node(to(d["low"]), to(d["high"]), to(d["chrom"]))
idx.build()
out = {"lib": os.environ.get("BIVX_LIB", "default"), "config": a.config}


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return round(ts[len(ts) // 2], 5)


for order in ("gen", "sorted"):
    if a.only and not a.only.startswith(order):
        continue
    perm = np.lexsort((d["qlow"], d["qchrom"])) if order == "sorted" else slice(None)
    c, lo, hi = to(d["qchrom"][perm]), to(d["qlow"][perm]), to(d["qhigh"][perm])
    Q = lo.numel()
    off = torch.empty(Q + 1, dtype=torch.int64, device=dev)
    idx.count_overlaps_device(lo, hi, c, offsets=off)
    H = int(off[-1].item())
    hits = torch.empty(H, dtype=torch.int32, device=dev)
    full = lambda: idx.query_device(lo, hi, off, hits, qchrom=c, sort_by_id=a.sort_ids)
    count = lambda: idx.count_overlaps_device(lo, hi, c, offsets=off)
    if a.only:
        fn = full if a.only.endswith("full") else count
        for _ in range(6):
            fn()
        torch.cuda.synchronize()
        out[a.only] = "ran 6"
    else:
        out[order] = {"full_ms": timed(full, a.reps), "count_ms": timed(count, a.reps), "H": H}
        out[order]["checksum"] = int(hits.to(torch.int64).sum().item()) ^ int(off.sum().item())
    del hits
print(json.dumps(out), flush=True)
