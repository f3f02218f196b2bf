#!/usr/bin/env python3
"""Randomized sessions: the pipelined kernels (BIVX_PIPE=2: every eligible batch) against k_query_fused (BIVX_PIPE=0) on
random indexes, batches, orders, capacities, launch limits and workgroup counts — offsets and ids must be identical,
and counts must equal the predicate's on a sample; and bivx_self_overlaps_dev against the general call on the same index. usage (on the GPU box): fuzz_pipe.py [sessions=200] [first seed=0]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from binary_amd import IntervalIndex  # noqa: E402

dev = torch.device("cuda:0")
to = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)
N, S0 = (int(sys.argv[1]) if len(sys.argv) > 1 else 200), (int(sys.argv[2]) if len(sys.argv) > 2 else 0)
KNOBS = ("BIVX_PIPE", "BIVX_MAX_TILES_PER_LAUNCH", "BIVX_PIPE_WGS")
bad = 0
for seed in range(S0, S0 + N):
    rng = np.random.default_rng(seed)
    big = os.environ.get("FUZZ_BIG") == "1"   # fewer, larger sessions: thousands of tiles, every workgroup busy
    n = int(10 ** (rng.uniform(5.5, 6.7) if big else rng.uniform(3, 5.7)))
    q = int(10 ** (rng.uniform(5.8, 6.5) if big else rng.uniform(3, 5.5)))
    nchrom = int(rng.integers(1, 31))
    L = int(10 ** (rng.uniform(7, 9) if big else rng.uniform(5, 8.3)))
    lmax = int(10 ** rng.uniform(1, 3.7))
    dense = rng.random() < 0.3
    if dense:                       # many ids per query: the regenerating kernel's territory when position-sorted
        L = max(int(n * lmax / rng.uniform(8, 40)), 1000)
    chrom = rng.integers(0, nchrom, n).astype(np.uint32)
    low = rng.integers(0, L, n).astype(np.uint32)
    high = (low + rng.integers(0, lmax + 1, n)).astype(np.uint32)
    if rng.random() < 0.4:          # an SV-like length spectrum: several length classes per chromosome, the longest not packed
        high = (low + np.exp(rng.uniform(np.log(10), np.log(10 ** rng.uniform(3, 6)), n))).astype(np.uint32)
    if rng.random() < 0.2:          # a few inverted records
        k = rng.permutation(n)[: n // 50]
        low[k], high[k] = high[k].copy(), low[k].copy()
    qc = rng.integers(0, nchrom + (1 if rng.random() < 0.3 else 0), q).astype(np.uint32)
    qlo = rng.integers(0, L, q).astype(np.uint32)
    qhi = (qlo + (rng.integers(0, lmax + 1, q) if rng.random() < 0.7 else 0)).astype(np.uint32)
    order = rng.choice(["generated", "sorted", "nearly"])
    if order != "generated":
        p = np.lexsort((qlo, qc))
        if order == "nearly":
            sw = rng.permutation(q)[: max(q // 60, 1)]
            p[sw] = p[np.roll(sw, 1)]
        qc, qlo, qhi = qc[p], qlo[p], qhi[p]
    # interval types (one segment range per chromosome and type) and a query for one type or all; a fused filter
    typ = rng.integers(1, 4, n).astype(np.uint8) if rng.random() < 0.3 else None
    qtype = int(rng.integers(0, 5)) if typ is not None else 0
    from binary_amd import capi
    fkind = int(rng.choice([capi.FILTER_SV2NL_DUP, capi.FILTER_SV2NL_INV, capi.FILTER_SV2NL_TRA])) if rng.random() < 0.3 else 0
    by_id = bool(rng.random() < 0.4)
    use_ws = bool(rng.random() < 0.3)
    knobs = {}
    if rng.random() < 0.3:
        knobs["BIVX_MAX_TILES_PER_LAUNCH"] = str(int(rng.integers(1, 40)))
    if rng.random() < 0.3:
        knobs["BIVX_PIPE_WGS"] = str(int(rng.integers(1, 700)))
    with IntervalIndex(0) as idx:
        idx.insert_node(low, high, chrom, svtype=typ)
        idx.build()
        d_qlo, d_qhi, d_qc = to(qlo), to(qhi), to(qc)
        flt = None
        if fkind:
            t_qaux = to(rng.integers(0, 12, q).astype(np.uint32))
            t_iaux = to(rng.integers(0, 12, n).astype(np.uint32))
            flt = IntervalIndex.device_filter(fkind, int(10 ** rng.uniform(2, 4.5)), bool(rng.random() < 0.5), t_qaux, t_iaux, svtype=qtype)
        elif qtype:
            flt = IntervalIndex.type_filter(qtype)
        os.environ["BIVX_PIPE"] = "0"
        off0 = torch.empty(q + 1, dtype=torch.int64, device=dev)   # (a one-id buffer: the offsets are complete all the same)
        idx.query_device(d_qlo, d_qhi, off0, torch.empty(1, dtype=torch.int32, device=dev), qchrom=d_qc, flt=flt)
        H = int(off0[-1].item())
        capm = rng.choice(["exact", "small", "zero"], p=[0.7, 0.2, 0.1])
        cap = H if capm == "exact" else (H // 3 if capm == "small" else 0)
        res = []
        for mode in ("2", "0"):
            for k in KNOBS:
                os.environ.pop(k, None)
            os.environ["BIVX_PIPE"] = mode
            os.environ.update(knobs)
            off = torch.full((q + 1,), -1, dtype=torch.int64, device=dev)
            hits = torch.full((max(cap, 1),), -1, dtype=torch.int32, device=dev)
            ws = torch.empty(idx.query_workspace_bytes(q), dtype=torch.uint8, device=dev) if use_ws else None
            name = idx.query_kernel_name(q, max(cap, 1), by_id, flt)
            idx.query_device(d_qlo, d_qhi, off, hits[:cap] if cap else hits[:0], workspace=ws, qchrom=d_qc, sort_by_id=by_id, flt=flt)
            idx.stream_status()
            res.append((off, hits, name))
        for k in KNOBS:
            os.environ.pop(k, None)
        # (a buffer that is too small and ascending ids: the ONE list that straddles the capacity may hold any of its
        #  ids below it — ordered in the stage it is cut after sorting, ordered behind a fill it is cut before; bivx.h)
        upto = cap
        if by_id and cap < H:
            o = res[1][0]
            upto = int(o[int(torch.searchsorted(o, torch.tensor([cap], device=dev), right=True).item()) - 1].item())
        ok = torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1][:upto], res[1][1][:upto]) and torch.equal(res[0][0], off0)
        # the predicate itself on a sample of queries
        sel = rng.integers(0, q, 40)
        cnt = (res[0][0][1:] - res[0][0][:-1]).cpu().numpy()
        for i in sel:
            m = (chrom == qc[i]) & (low <= qhi[i]) & (high >= qlo[i])
            if qtype:
                m &= typ == qtype
            ok = ok and (bool(fkind) or int(m.sum()) == int(cnt[i]))   # (the filters' own predicates: tests/test_gpu_pipe_ms.py)
        # the index overlapped with itself: bivx_self_overlaps_dev (BIVX_PIPE=2: its fast path whenever the index allows
        # it) against the general call with the appended columns as the batch (BIVX_PIPE=0)
        self_ok = True
        if (n <= 400_000 or big) and typ is None:
            d_lo, d_hi, d_c = to(low), to(high), to(chrom)
            os.environ["BIVX_PIPE"] = "0"
            soff0 = idx.count_overlaps_device(d_lo, d_hi, d_c)
            SH = int(soff0[-1].item())
            if SH < 400_000_000:
                sh0 = torch.full((max(SH, 1),), -1, dtype=torch.int32, device=dev)
                so0 = torch.empty(n + 1, dtype=torch.int64, device=dev)
                idx.query_device(d_lo, d_hi, so0, sh0[:SH], qchrom=d_c, sort_by_id=by_id)
                os.environ["BIVX_PIPE"] = "2"
                sh1 = torch.full((max(SH, 1),), -1, dtype=torch.int32, device=dev)
                so1 = torch.full((n + 1,), -1, dtype=torch.int64, device=dev)
                idx.self_overlaps_device(so1, sh1[:SH], sort_by_id=by_id)
                idx.stream_status()
                self_ok = torch.equal(so0, so1) and torch.equal(sh0, sh1) and torch.equal(so0, soff0)
            os.environ.pop("BIVX_PIPE", None)
        ok = ok and self_ok
        st = idx.stats()
        ok = ok and st["prefix_timeouts"] == 0
    tag = "ok " if ok else "BAD"
    bad += 0 if ok else 1
    print(f"{tag} seed {seed}: n={n} q={q} chroms={nchrom} L={L} lmax={lmax} H={H} ({H / q:.1f}/query) {order} by_id={by_id} cap={capm} "
          f"ws={use_ws} types={typ is not None}/{qtype} filter={fkind} segs={st['n_segments']} {knobs} -> {res[0][2]}", flush=True)
print(f"{N} sessions, {bad} bad")
sys.exit(1 if bad else 0)
