#!/usr/bin/env python3
"""Times bivx_build (the batch replacement of the reference's per-record insert_node loop, rb_tree.hpp:111-117 +
interval_tree.hpp:230-260) on a BASELINE config's interval set: device-resident append, then `--reps` rebuilds.

    python3 tools/build_bench.py [--config 3] [--reps 20] [--typed]

Prints one JSON line: wall time of bivx_build as the library reports it (host clock around the whole call, syncs
included) and the build roofline: algorithmic bytes 34 N
(12 N in: chrom, low, high; out: se 8 N + id 4 N + rec 8 N + directory ~2 N) over the HBM peak.
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split (tools/kstats.py prints it)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binary_amd import IntervalIndex, synth  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--config", type=int, default=3, choices=(2, 3, 5))
p.add_argument("--reps", type=int, default=20)
p.add_argument("--typed", action="store_true", help="label the intervals with 3 svtypes (sv2nl's one typed index)")
p.add_argument("--lmax", type=int, default=1000)
a = p.parse_args()

dev = torch.device("cuda:0")
to = lambda x: torch.from_numpy(np.ascontiguousarray(x).view(np.int32)).to(dev)
if a.config == 2:
    low, high = synth.gen_intervals(1_000_000, int(synth.HG38_LENGTHS[0]), a.lmax, 0)
    chrom = None
else:
    d = synth.gen_genome(10_000_000 if a.config == 3 else 50_000_000, 0, a.lmax)
    low, high, chrom = d["low"], d["high"], d["chrom"]
N = int(low.size)
d_lo, d_hi = to(low), to(high)
d_c = None if chrom is None else to(chrom)
d_t = torch.from_numpy((np.arange(N) % 3 + 1).astype(np.uint8)).to(dev) if a.typed else None

import time  # noqa: E402

wall, app, both = [], [], []
with IntervalIndex(0) as idx:
    for r in range(a.reps + 2):
        idx.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        idx.insert_node(d_lo, d_hi, d_c, svtype=d_t)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        idx.build()
        t2 = time.perf_counter()
        if r >= 2:  # the first builds allocate the index's pooled blocks
            wall.append(idx.stats()["build_ms"])
            app.append((t1 - t0) * 1e3)
            both.append((t2 - t0) * 1e3)
    st = idx.stats()
b_alg = 34 * N
ms = float(np.median(wall))
print(json.dumps({"commit": os.environ.get("BIVX_GIT_REV"), "config": a.config, "intervals": N, "typed": bool(a.typed), "reps": a.reps,
                  "build_ms_median": ms, "build_ms_min": float(np.min(wall)), "build_ms_max": float(np.max(wall)),
                  "append_ms_median": float(np.median(app)), "append_plus_build_ms_median": float(np.median(both)),
                  "append_note": "bivx_append_dev of the device-resident columns + a device synchronisation, host clock (Python call "
                                 "overhead included); untyped appends also leave the build's statistics behind",
                  "intervals_per_s": N / ms * 1e3, "algorithmic_bytes": b_alg,
                  "roofline": {"bound": "hbm", "achieved_GBs": b_alg / ms / 1e6, "peak_GBs": 8000.0,
                               "frac": b_alg / ms / 1e6 / 8000.0},
                  "segments": st["n_segments"], "cells": st["n_cells"], "index_bytes": st["index_bytes"]}))
