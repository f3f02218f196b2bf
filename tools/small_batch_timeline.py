#!/usr/bin/env python3
"""Where the fixed cost of a SMALL pipelined launch goes: wall-clock stamps (s_memrealtime, 100 MHz)
of every workgroup's life in k_query_pipe — entry, descriptors staged, first ticket, first probe issued, worker 0's last
slice out, service wavefront gone — relative to the first workgroup's entry, for one launch of config 2 (1 M point queries)
or of rank 0's N = 8 shard of config 4. Needs a library built with -DBIVX_STAMPS:
    tools/build_variant.sh stamps "-DBIVX_STAMPS" query_pipe query_fused
    BIVX_LIB=binary_amd/libbivx.so.stamps python tools/small_batch_timeline.py [queries=1000000]
The stamped build's own run time is not quoted as a result."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binary_amd import IntervalIndex, synth, capi  # noqa: E402

Q = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda:0")
L = int(synth.HG38_LENGTHS[0])
lo, hi = synth.gen_intervals(1_000_000, L, 1000, 0)
ql, qh = synth.gen_point_queries(Q, L, 0)
to = lambda a: torch.from_numpy(a.view(np.int32)).to(dev)
idx = IntervalIndex(0)
idx.insert_node(lo, hi)
idx.build()
dql, dqh = to(ql), to(qh)
off = torch.empty(Q + 1, dtype=torch.int64, device=dev)
hits = torch.empty(3 * Q + 1024, dtype=torch.int32, device=dev)
os.environ["BIVX_PIPE"] = "2"
for _ in range(20):
    idx.query_device(dql, dqh, off, hits)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
idx.query_device(dql, dqh, off, hits)
e1.record()
torch.cuda.synchronize()
n = 1024 * 8
buf = (C.c_ulonglong * n)()
assert capi.load().bivx_debug_wgstamps(buf, n) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8).astype(np.int64)
st = st[st[:, 0] > 0]
t0 = st[:, 0].min()
us = (st[:, :6] - t0) / 100.0
print(f"one launch of {Q} queries ({(Q + 959) // 960} tiles of 960) on {len(st)} workgroups; step (both launches, HIP events) "
      f"{e0.elapsed_time(e1) * 1e3:.1f} us [stamped build]")
names = ["workgroup enters", "descriptors staged (after the barrier)", "first ticket seen by worker 0",
         "first queries + directory probe issued", "worker 0 leaves (its last slice is out)", "service wavefront leaves"]
for k, nm in enumerate(names):
    v = us[:, k]
    print(f"  {nm:44s} first {v.min():6.2f}  median {np.median(v):6.2f}  p90 {np.percentile(v, 90):6.2f}  last {v.max():6.2f} us")
it = st[:, 6]
print(f"  tiles per workgroup: min {it.min()}  median {int(np.median(it))}  max {it.max()}")
life = us[:, 4] - us[:, 3]
per = life / np.maximum(it, 1)
print(f"  worker 0's loop (first probe .. leaves): median {np.median(life):.2f} us = {np.median(per):.2f} us per tile drawn")
print(f"  kernel's end (last service wavefront) at {us[:, 5].max():.2f} us; the last worker leaves at {us[:, 4].max():.2f}, the first at {us[:, 4].min():.2f}")
