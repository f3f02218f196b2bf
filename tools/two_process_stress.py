#!/usr/bin/env python3
"""Two PROCESSES on one GPU, each with an index of its own, both launching the pipelined kernels at the same time (what
`bench.py --gpus 2` rehearsed on a one-GPU box does, and what two tools sharing a card do): persistent workgroups of two
launches compete for the CUs, a launch may run with only part of its grid resident for a long time. Every call must finish
promptly with the right totals. usage: two_process_stress.py [intervals=5e6] [rounds=20]"""
import multiprocessing as mp
import os
import sys
import time

import numpy as np


def work(rank, n, rounds, q):
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from binary_amd import IntervalIndex, synth
    dev = torch.device("cuda:0")
    to = lambda x: torch.from_numpy(np.ascontiguousarray(x).view(np.int32)).to(dev)
    d = synth.gen_genome(n, n, 1000)
    idx = IntervalIndex(0)
    idx.insert_node(to(d["low"]), to(d["high"]), to(d["chrom"]))
    idx.build()
    ql, qh, qc = to(d["qlow"]), to(d["qhigh"]), to(d["qchrom"])
    off = torch.empty(n + 1, dtype=torch.int64, device=dev)
    worst, total = 0.0, None
    for r in range(rounds):
        t0 = time.perf_counter()
        idx.count_overlaps_device(ql, qh, qc, offsets=off)
        idx.stream_status()
        h = int(off[-1].item())
        if total is None:
            total = h
            hits = torch.empty(h, dtype=torch.int32, device=dev)
        assert h == total, (rank, r, h, total)
        idx.query_device(ql, qh, off, hits, qchrom=qc)
        idx.stream_status()
        assert int(off[-1].item()) == total
        off_s = torch.empty(n + 1, dtype=torch.int64, device=dev)
        idx.self_overlaps_device(off_s, hits[:0])
        idx.stream_status()
        worst = max(worst, time.perf_counter() - t0)
    q.put((rank, total, worst))


if __name__ == "__main__":
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 5_000_000
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=work, args=(r, n, rounds, q)) for r in range(2)]
    t0 = time.perf_counter()
    for p in ps:
        p.start()
    for p in ps:
        p.join(timeout=600)
    res = sorted(q.get() for _ in range(sum(p.exitcode == 0 for p in ps)))
    print({"exit codes": [p.exitcode for p in ps], "results (rank, ids, slowest round s)": res,
           "wall_s": round(time.perf_counter() - t0, 1)})
    sys.exit(0 if all(p.exitcode == 0 for p in ps) and all(r[2] < 5.0 for r in res) else 1)
