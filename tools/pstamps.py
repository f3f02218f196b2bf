#!/usr/bin/env python3
"""Diagnostic: builds libbivx.so with -DBIVX_STAMPS, runs the pipelined single-pass kernel (query_pipe.hip) on config 3
and prints where an iteration of a workgroup spends its time (shares; the stamped build's run time is not quoted)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SORTED = "--sorted" in sys.argv
subprocess.check_call(["make", "-C", os.path.join(ROOT, "binary_amd", "csrc"), "-s", "clean"])
subprocess.check_call(["make", "-C", os.path.join(ROOT, "binary_amd", "csrc"), "-s", "-j8", "EXTRA=-DBIVX_STAMPS"])
from binary_amd import IntervalIndex, synth, capi  # noqa: E402

dev = torch.device("cuda:0")
d = synth.gen_genome(10_000_000, 10_000_000, 1000)
ch, lo, hi, qc, ql, qh = (d[k] for k in ("chrom", "low", "high", "qchrom", "qlow", "qhigh"))
if SORTED:
    o = np.lexsort((ql, qc))
    qc, ql, qh = qc[o], ql[o], qh[o]
to = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)
idx = IntervalIndex(0)
idx.insert_node(lo, hi, ch)
idx.build()
Q = ql.size
off = torch.empty(Q + 1, dtype=torch.int64, device=dev)
hits = torch.empty(40_000_000, dtype=torch.int32, device=dev)
dql, dqh, dqc = to(ql), to(qh), to(qc)
for _ in range(5):
    idx.query_device(dql, dqh, off, hits, qchrom=dqc)
torch.cuda.synchronize()
n = 1024 * 8
buf = (C.c_ulonglong * n)()
assert capi.load().bivx_debug_pstamps(buf, n) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8).astype(np.int64)
st = st[st[:, 0] > 0]
us = (st[:, :7] - st[:, 0].min()) / 100.0
names = ["iteration start", "counted (wave 0)", "barrier A", "published + swept (wave 0)", "barrier B", "pending tile out",
         "ids laid out"]
dd = np.diff(us, axis=1)
print(f"tiles stamped: {len(st)}")
for k in range(6):
    print(f"  {names[k]:>28s} -> {names[k+1]:<28s} median {np.median(dd[:, k]):6.2f}  p90 {np.percentile(dd[:, k], 90):6.2f}")
print(f"  whole iteration: median {np.median(us[:, 6] - us[:, 0]):6.2f}  p90 {np.percentile(us[:, 6] - us[:, 0], 90):6.2f}")
subprocess.check_call(["make", "-C", os.path.join(ROOT, "binary_amd", "csrc"), "-s", "clean"])
subprocess.check_call(["make", "-C", os.path.join(ROOT, "binary_amd", "csrc"), "-s", "-j8"])
