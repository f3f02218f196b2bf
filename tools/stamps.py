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
extra = " ".join(a for a in sys.argv[1:] if a not in ("--sorted", "--dense"))
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
if SORTED:
    o = np.argsort(ql, kind="stable")
    ql, qh = ql[o], qh[o]
to = lambda a: torch.from_numpy(a.view(np.int32)).to(dev)
idx = IntervalIndex(0)
idx.insert_node(lo, hi)
idx.build()
Q = ql.size
off = torch.empty(Q + 1, dtype=torch.int64, device=dev)
hits = torch.empty(30_000_000 if DENSE else 3_000_000, dtype=torch.int32, device=dev)
ws = torch.empty(idx.query_workspace_bytes(Q), dtype=torch.uint8, device=dev)
dql, dqh = to(ql), to(qh)
for _ in range(20):
    idx.query_device(dql, dqh, off, hits, ws)
torch.cuda.synchronize()
n = 1024 * 8
buf = (C.c_ulonglong * n)()
assert capi.load().bivx_debug_stamps(buf, n) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8).astype(np.int64)
st = st[st[:, 0] > 0]
t0 = st[:, 0].min()
us = (st[:, :7] - t0) / 100.0  # 100 MHz -> microseconds
names = ["entry", "ticket+LDS", "phase1 done (wave0)", "scan done/publish", "prefix known", "barrier", "end"]
print(f"tiles stamped: {len(st)}; kernel span {us[:, 6].max():.1f} us")
for k, nm in enumerate(names):
    print(f"  {nm:24s} min {us[:, k].min():7.2f}  median {np.median(us[:, k]):7.2f}  max {us[:, k].max():7.2f}")
d = np.diff(us, axis=1)
for k in range(6):
    print(f"  segment {names[k]:>22s} -> {names[k+1]:<22s} median {np.median(d[:, k]):7.2f}  p90 {np.percentile(d[:, k], 90):7.2f}  max {d[:, k].max():7.2f}")
order = np.argsort(st[:, 7])
print("  ticket order vs entry time (first 8 tickets):", us[order[:8], 0].round(2).tolist())
subprocess.check_call(["make", "-C", os.path.join(ROOT, "binary_amd", "csrc"), "-s", "clean"])
subprocess.check_call(["make", "-C", os.path.join(ROOT, "binary_amd", "csrc"), "-s", "-j8"])
