#!/usr/bin/env python3
"""Latency of the host-pointer entry points called with ONE query at a time — what unmodified reference-style
code does (one find_overlaps per NL record, mapper.hpp:205-231) when it only swaps the header."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binary_amd import IntervalIndex, capi, synth  # noqa: E402

L = capi.load()
lo, hi = synth.gen_intervals(100_000, 50_000_000, 1000, 0)
idx = IntervalIndex(0)
idx.insert_node(lo, hi)
idx.build()
ql, qh = synth.gen_range_queries(2000, 50_000_000, 1000, 0)
p = lambda a: a.ctypes.data_as(C.c_void_p)
off = np.zeros(2, np.uint64)
for name in ("find_overlaps", "any"):
    for rep in range(2):
        t0 = time.perf_counter()
        for i in range(ql.size):
            a, b = ql[i:i + 1], qh[i:i + 1]
            if name == "find_overlaps":
                hp = C.POINTER(C.c_uint32)()
                capi.check(L.bivx_find_overlaps(idx._h, None, p(a), p(b), 1, None, 1, p(off), C.byref(hp)))
                L.bivx_free(hp)
            else:
                first = np.zeros(1, np.uint32)
                capi.check(L.bivx_any(idx._h, None, p(a), p(b), 1, p(first)))
        dt = (time.perf_counter() - t0) / ql.size
    print(f"bivx_{name}: {dt * 1e6:.1f} us per single-query call")
