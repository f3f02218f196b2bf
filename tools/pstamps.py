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
PREBUILT = bool(os.environ.get("BIVX_LIB"))   # a library built with -DBIVX_STAMPS by tools/build_variant.sh: nothing is compiled here
if not PREBUILT:
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "binary_amd", "csrc"), "-s", "clean"])
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "binary_amd", "csrc"), "-s", "-j8", "EXTRA=-DBIVX_STAMPS"])
os.environ["BIVX_PIPE"] = "2"
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
# stamps of worker wavefront 0, per tile: 0 counting starts, 1 total reported, 2 next ticket seen + its queries issued,
# 3 (of the tile flushed one iteration later) pending slice out, 4 ids laid out
us = (st[:, :5] - st[:, 0].min()) / 100.0
print(f"tiles stamped: {len(st)}")
for a, b, nm in ((0, 1, "counting"), (1, 2, "wait for the next ticket + issue its queries"),
                 (2, 3, "the slice of two iterations ago goes out (incl. wait for its base)"), (3, 4, "lay the new slice's ids out")):
    dd = us[:, b] - us[:, a]
    dd = dd[(dd > 0) & (dd < 1000)]
    print(f"  {nm:68s} median {np.median(dd):6.2f}  p90 {np.percentile(dd, 90):6.2f}")
dd = us[:, 4] - us[:, 0]
dd = dd[(dd > 0) & (dd < 1000)]
print(f"  whole iteration (worker 0): median {np.median(dd):6.2f}  p90 {np.percentile(dd, 90):6.2f}")
if not PREBUILT:
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "binary_amd", "csrc"), "-s", "clean"])
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "binary_amd", "csrc"), "-s", "-j8"])
