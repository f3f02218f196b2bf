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
SKEWED = next((float(a.split("=")[1]) for a in sys.argv if a.startswith("--skewed=")), 0)  # tools/skewed_bench.py's workload, MAXLEN
extra = " ".join(a for a in sys.argv[1:] if a not in ("--sorted", "--dense", "--config3") and not a.startswith("--skewed="))
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
if SKEWED:
    rng = np.random.default_rng(5)
    G = 248_956_422
    lo = rng.integers(0, G - 10_000_001, size=1_000_000).astype(np.uint32)
    hi = lo + np.exp(rng.uniform(np.log(50), np.log(SKEWED), size=lo.size)).astype(np.uint32)
    ql = qh = rng.integers(0, G, size=1_000_000).astype(np.uint32)
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
hits = torch.empty(80_000_000 if SKEWED else 40_000_000 if (DENSE or CONFIG3) else 3_000_000, dtype=torch.int32, device=dev)
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
names = ["tile start", "ticket + descriptors", "counted", "scanned (wave 0)", "sweeps done (wave 0)", "barrier B", "tile end"]
print("segments", idx.stats()["n_segments"], "kernel", idx.last_kernel_name() if hasattr(idx, "last_kernel_name") else "")
ids = (st[:, 8] - st[:, 5]) / 100.0
print(f"  wave 0: barrier B -> its ids out: median {np.median(ids):7.2f}  p90 {np.percentile(ids, 90):7.2f}  max {ids.max():7.2f}")
print(f"tiles stamped: {len(st)}; span {us[:, 6].max():.1f} us")
for k, nm in enumerate(names):
    print(f"  {nm:24s} min {us[:, k].min():7.2f}  median {np.median(us[:, k]):7.2f}  max {us[:, k].max():7.2f}")
d = np.diff(us, axis=1)
for k in range(6):
    print(f"  segment {names[k]:>22s} -> {names[k+1]:<22s} median {np.median(d[:, k]):7.2f}  p90 {np.percentile(d[:, k], 90):7.2f}  max {d[:, k].max():7.2f}")
print(f"  whole iteration: median {np.median(us[:, 6] - us[:, 0]):7.2f}  p90 {np.percentile(us[:, 6] - us[:, 0], 90):7.2f}")
subprocess.check_call(["make", "-C", os.path.join(ROOT, "binary_amd", "csrc"), "-s", "clean"])
subprocess.check_call(["make", "-C", os.path.join(ROOT, "binary_amd", "csrc"), "-s", "-j8"])
