#!/usr/bin/env python3
"""Splits the single-pass kernel's time by what it is asked to produce, on position-sorted and generation-order
batches: count only (zero capacity: phase 1 + prefix + offsets), full CSR, unordered output. Diagnostic."""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from binary_amd import IntervalIndex, synth  # noqa: E402

dev = torch.device("cuda:0")
to = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def run(name, d, self_overlap=False):
    idx = IntervalIndex(0)
    idx.insert_node(d["low"], d["high"], d["chrom"])
    idx.build()
    out = {"name": name}
    qc, ql, qh = (d["chrom"], d["low"], d["high"]) if self_overlap else (d["qchrom"], d["qlow"], d["qhigh"])
    for order in ("generated", "sorted"):
        perm = np.lexsort((ql, qc)) if order == "sorted" else np.arange(ql.size)
        c, lo, hi = to(qc[perm]), to(ql[perm]), to(qh[perm])
        Q = lo.numel()
        off = torch.empty(Q + 1, dtype=torch.int64, device=dev)
        idx.count_overlaps_device(lo, hi, c, offsets=off)
        H = int(off[-1].item())
        hits = torch.empty(H, dtype=torch.int32, device=dev)
        beg = torch.empty(Q, dtype=torch.int64, device=dev)
        cnt = torch.empty(Q, dtype=torch.int32, device=dev)
        tot = torch.zeros(1, dtype=torch.int64, device=dev)
        out[order] = {
            "H": H,
            "count_only_ms": timed(lambda: idx.count_overlaps_device(lo, hi, c, offsets=off)),
            "full_ms": timed(lambda: idx.query_device(lo, hi, off, hits, qchrom=c)),
            "full_ascending_id_ms": timed(lambda: idx.query_device(lo, hi, off, hits, qchrom=c, sort_by_id=True)),
            "unordered_ms": timed(lambda: idx.query_device_unordered(lo, hi, beg, cnt, hits, tot, qchrom=c)),
            "unordered_count_only_ms": timed(lambda: idx.query_device_unordered(lo, hi, beg, cnt, hits[:0], tot, qchrom=c)),
        }
        del hits
    idx.close()
    print(json.dumps(out), flush=True)


which = sys.argv[1:] or ["3", "5"]
if "2" in which:
    L = int(synth.HG38_LENGTHS[0])
    lo, hi = synth.gen_intervals(1_000_000, L, 1000, 0)
    ql, qh = synth.gen_point_queries(1_000_000, L, 0)
    z = np.zeros(1_000_000, np.uint32)
    run("config2", dict(chrom=z, low=lo, high=hi, qchrom=z, qlow=ql, qhigh=qh))
if "3" in which:
    run("config3", synth.gen_genome(10_000_000, 10_000_000, 1000))
if "5" in which:
    run("config5", synth.gen_genome(50_000_000, 0, 1000), self_overlap=True)
