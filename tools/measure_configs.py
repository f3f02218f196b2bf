#!/usr/bin/env python3
"""Side measurements quoted in DESIGN.md (not the headline bench): BASELINE.json configs 2 and 3 on one GPU,
random vs position-sorted queries, single-pass vs two-pass, build time, and the PCIe-inclusive rate of the
host-pointer entry points. Prints one JSON object."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from binary_amd import IntervalIndex, synth  # noqa: E402


def timed(fn, reps, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def run_config(name, data, reps):
    dev = torch.device("cuda:0")
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)
    out = {"name": name, "intervals": int(data["low"].size), "queries": int(data["qlow"].size)}
    idx = IntervalIndex(0)
    t0 = time.perf_counter()
    idx.insert_node(data["low"], data["high"], data["chrom"])
    idx.build()
    out["append_plus_build_host_ms"] = (time.perf_counter() - t0) * 1e3
    st = idx.stats()
    out["build_ms"] = st["build_ms"]
    out["segments"] = st["n_segments"]
    out["index_bytes"] = st["index_bytes"]
    Q = data["qlow"].size
    for order in ("random", "sorted"):
        if order == "sorted":
            perm = np.lexsort((data["qlow"], data["qchrom"]))
        else:
            perm = np.arange(Q)
        qc, ql, qh = to(data["qchrom"][perm]), to(data["qlow"][perm]), to(data["qhigh"][perm])
        off = torch.empty(Q + 1, dtype=torch.int64, device=dev)
        qws = torch.empty(idx.query_workspace_bytes(Q), dtype=torch.uint8, device=dev)
        idx.count_overlaps_device(ql, qh, qc, offsets=off)
        H = int(off[-1].item())
        hits = torch.empty(max(H, 1), dtype=torch.int32, device=dev)
        ms1 = timed(lambda: idx.query_device(ql, qh, off, hits, qws, qchrom=qc), reps)
        ms2 = timed(lambda: (idx.count_overlaps_device(ql, qh, qc, offsets=off),
                             idx.fill_overlaps_device(ql, qh, off, hits, qchrom=qc)), reps)
        ms3 = timed(lambda: idx.query_device(ql, qh, off, hits, qws, qchrom=qc, sort_by_id=True), reps)
        beg = torch.empty(Q, dtype=torch.int64, device=dev)
        cnt = torch.empty(Q, dtype=torch.int32, device=dev)
        tot = torch.zeros(1, dtype=torch.int64, device=dev)
        ms4 = timed(lambda: idx.query_device_unordered(ql, qh, beg, cnt, hits, tot, qchrom=qc), reps)
        assert int(tot.item()) == H
        b_alg = 8 * data["low"].size + 8 * Q + 8 * (Q + 1) + 4 * H
        out[order] = {"hits": H, "single_pass_ms": ms1, "two_pass_ms": ms2, "single_pass_sorted_ids_ms": ms3,
                      "unordered_begin_count_ms": ms4, "unordered_gqps": Q / ms4 / 1e6,
                      "single_pass_gqps": Q / ms1 / 1e6, "algorithmic_gbs": b_alg / ms1 / 1e6,
                      "algorithmic_frac_of_8tbs": b_alg / ms1 / 1e6 / 8000.0}
    # host-pointer API: upload queries, count, download offsets, fill, download hits (PCIe inclusive)
    t0 = time.perf_counter()
    off_h, hits_h = idx.find_overlaps(data["qlow"], data["qhigh"], data["qchrom"], sort_by_id=False)
    out["host_api_ms"] = (time.perf_counter() - t0) * 1e3
    out["host_api_gqps"] = Q / out["host_api_ms"] / 1e6
    idx.close()
    return out


def main():
    res = []
    L = int(synth.HG38_LENGTHS[0])
    lo, hi = synth.gen_intervals(1_000_000, L, 1000, 0)
    z = np.zeros(1_000_000, np.uint32)
    for kind, (ql, qh) in (("point", synth.gen_point_queries(1_000_000, L, 0)),
                           ("range", synth.gen_range_queries(1_000_000, L, 1000, 0))):
        res.append(run_config(f"config2-{kind}: 1 chrom, 1M x 1M", dict(chrom=z, low=lo, high=hi, qchrom=z, qlow=ql, qhigh=qh), 50))
    res.append(run_config("config3: 24 chroms, 10M x 10M range", synth.gen_genome(10_000_000, 10_000_000, 1000), 10))
    print(json.dumps({"device": torch.cuda.get_device_name(0), "results": res}, indent=1))


if __name__ == "__main__":
    main()
