#!/usr/bin/env python3
"""Diagnostic: builds libbivx.so with -DBIVX_STAMPS, runs the single-pass kernel on the bench workload and
prints where a tile spends its time (shares; the stamped build's run time itself is not quoted anywhere)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SORTED = "--sorted" in sys.argv
DENSE = "--dense" in sys.argv  # config-5 density: self-overlap, ~17 ids per query
CONFIG3 = "--config3" in sys.argv  # 24 chromosomes, 10 M x 10 M range queries (the stamps of the last 1024 tiles survive)
extra = " ".join(a for a in sys.argv[1:] if a not in ("--sorted", "--dense", "--config3"))
subprocess.check_call(["make", "-C", os.path.join(ROOT, "binary_amd", "csrc"), "-s", "clean"])
subprocess.check_call(["make", "-C", os.path.join(ROOT, "binary_amd", "csrc"), "-s", "-j8", f"EXTRA=-DBIVX_STAMPS {extra}"])
from binary_amd import IntervalIndex, synth, capi  # noqa: E402

dev = torch.device("cuda:0")
L = int(synth.HG38_LENGTHS[0])
lo, hi = synth.gen_intervals(1_000_000, L, 1000, 0)
ql, qh = synth.gen_point_queries(1_000_000, L, 0)
if DENSE:
    lo, hi = synth.gen_intervals(1_000_000, 62_000_000, 1000, 0)
    ql, qh = lo.copy(), hi.copy()
ch = qc = None
if CONFIG3:
    d = synth.gen_genome(10_000_000, 10_000_000, 1000)
    ch, lo, hi, qc, ql, qh = (d[k] for k in ("chrom", "low", "high", "qchrom", "qlow", "qhigh"))
if SORTED:
    o = np.argsort(ql, kind="stable") if qc is None else np.lexsort((ql, qc))
    ql, qh = ql[o], qh[o]
    qc = None if qc is None else qc[o]
to = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)
idx = IntervalIndex(0)
idx.insert_node(lo, hi, ch)
idx.build()
Q = ql.size
off = torch.empty(Q + 1, dtype=torch.int64, device=dev)
hits = torch.empty(40_000_000 if (DENSE or CONFIG3) else 3_000_000, dtype=torch.int32, device=dev)
dql, dqh, dqc = to(ql), to(qh), to(qc)
for _ in range(5):
    idx.query_device(dql, dqh, off, hits, qchrom=dqc)
torch.cuda.synchronize()
n = 1024 * 12
buf = (C.c_ulonglong * n)()
assert capi.load().bivx_debug_stamps(buf, n) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 12).astype(np.int64)
st = st[st[:, 0] > 0]
t0 = st[:, 0].min()
us = (st[:, :7] - t0) / 100.0  # 100 MHz -> microseconds
names = ["iteration start", "counted", "ids staged", "published (wave 0)", "sweeps done (wave 0)", "barrier B", "iteration end"]
print(f"tiles stamped: {len(st)}; span {us[:, 6].max():.1f} us")
for k, nm in enumerate(names):
    print(f"  {nm:24s} min {us[:, k].min():7.2f}  median {np.median(us[:, k]):7.2f}  max {us[:, k].max():7.2f}")
d = np.diff(us, axis=1)
for k in range(6):
    print(f"  segment {names[k]:>22s} -> {names[k+1]:<22s} median {np.median(d[:, k]):7.2f}  p90 {np.percentile(d[:, k], 90):7.2f}  max {d[:, k].max():7.2f}")
print(f"  whole iteration: median {np.median(us[:, 6] - us[:, 0]):7.2f}  p90 {np.percentile(us[:, 6] - us[:, 0], 90):7.2f}")
subprocess.check_call(["make", "-C", os.path.join(ROOT, "binary_amd", "csrc"), "-s", "clean"])
subprocess.check_call(["make", "-C", os.path.join(ROOT, "binary_amd", "csrc"), "-s", "-j8"])
